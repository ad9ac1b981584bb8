/* oracle/snappy_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference's Snappy raw format path:
 *   compress   : snappy::RawCompress (algos/snappy/snappy.cc:2494-2666) -> Compress(Source*,Sink*) (:1749-1823)
 *                -> per <=64 KiB fragment AOCL_CompressFragment (:846-1046), EmitLiteral (:435-476),
 *                EmitCopy/EmitCopyAtMost64 (:478-568), HashBytes (:152-158), CalculateTableSize (:619-632),
 *                MaxCompressedLength (:160-182), Varint::Encode32 (snappy-stubs-internal.h:462).
 *   decompress : snappy::RawUncompress (:2271-2390) -> SnappyDecompressor::ReadUncompressedLength (:1202-1223),
 *                DecompressAllTags (:1466-1570), RefillTag (:1598-1653), SnappyArrayWriter (:2131-2209).
 * The probe schedule is written exactly as the kernel uses it: the 16 unrolled probes of :906-934 are the
 * first 16 steps of the skip recurrence of :935-974 (skip 32 -> 48, one byte per step), and their missing
 * end-of-input test is implied by the `ip_limit - ip >= 16` guard, so one loop with the test on every probe
 * visits the same positions and performs the same table writes.
 */
#include "oracle.h"
#include <string.h>
#include <stdlib.h>

enum { SN_BLOCK = 65536, SN_MAX_TABLE = 1 << 14, SN_MIN_TABLE = 1 << 8, SN_MARGIN = 15 };

static uint32_t s_rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

size_t orc_snappy_max_compressed_length(size_t n) { return 32 + n + n / 6; }

static int log2floor(uint32_t x) { int r = -1; while (x) { r++; x >>= 1; } return r; }

static uint32_t table_size_for(uint32_t n)
{   /* snappy.cc:619-632 */
    if (n > SN_MAX_TABLE) return SN_MAX_TABLE;
    if (n < SN_MIN_TABLE) return SN_MIN_TABLE;
    return 2u << log2floor(n - 1);
}

static uint8_t* emit_literal(uint8_t* op, const uint8_t* lit, uint32_t len)
{   /* snappy.cc:435-476 */
    const uint32_t n = len - 1;
    if (n < 60) *op++ = (uint8_t)(n << 2);
    else {
        const int count = (log2floor(n) >> 3) + 1;
        *op++ = (uint8_t)((59 + count) << 2);
        for (int i = 0; i < count; i++) *op++ = (uint8_t)(n >> (8 * i));
    }
    memcpy(op, lit, len);
    return op + len;
}

static uint8_t* emit_copy_le64(uint8_t* op, uint32_t offset, uint32_t len)
{   /* snappy.cc:478-505 */
    if (len < 12 && offset < 2048) {
        *op++ = (uint8_t)(1 | ((len - 4) << 2) | ((offset >> 8) << 5));
        *op++ = (uint8_t)offset;
    } else {
        *op++ = (uint8_t)(2 | ((len - 1) << 2));
        *op++ = (uint8_t)offset;
        *op++ = (uint8_t)(offset >> 8);
    }
    return op;
}

static uint8_t* emit_copy(uint8_t* op, uint32_t offset, uint32_t len)
{   /* snappy.cc:507-568 */
    if (len >= 12) {
        while (len >= 68) { op = emit_copy_le64(op, offset, 64); len -= 64; }
        if (len > 64) { op = emit_copy_le64(op, offset, 60); len -= 60; }
    }
    return emit_copy_le64(op, offset, len);
}

static uint8_t* compress_fragment(const uint8_t* in, uint32_t n, uint8_t* op, uint16_t* table, uint32_t tsize)
{
    const int shift = 32 - log2floor(tsize);
    uint32_t ip = 0;
    memset(table, 0, tsize * sizeof(uint16_t));                       /* snappy.cc:651-657 */
    if (n >= SN_MARGIN) {
        const uint32_t ip_limit = n - SN_MARGIN;
        for (;;) {
            const uint32_t next_emit = ip++;
            uint32_t skip = 32, cand;
            for (;;) {                                                /* snappy.cc:906-974 */
                const uint32_t bbhl = skip >> 5;
                skip += bbhl;
                const uint32_t next_ip = ip + bbhl;
                if (next_ip > ip_limit) { ip = next_emit; goto remainder; }
                const uint32_t h = (s_rd32(in + ip) * 0x1e35a7bdu) >> shift;
                cand = table[h];
                table[h] = (uint16_t)ip;
                if (s_rd32(in + cand) == s_rd32(in + ip)) break;
                ip = next_ip;
            }
            op = emit_literal(op, in + next_emit, ip - next_emit);    /* snappy.cc:979 */
            for (;;) {                                                /* emit_match loop, snappy.cc:990-1032 */
                uint32_t matched = 4;
                while (ip + matched < n && in[cand + matched] == in[ip + matched]) matched++;
                op = emit_copy(op, ip - cand, matched);
                ip += matched;
                if (ip >= ip_limit) goto remainder;
                table[(s_rd32(in + ip - 1) * 0x1e35a7bdu) >> shift] = (uint16_t)(ip - 1);
                const uint32_t h = (s_rd32(in + ip) * 0x1e35a7bdu) >> shift;
                cand = table[h];
                table[h] = (uint16_t)ip;
                if (s_rd32(in + cand) != s_rd32(in + ip)) break;
            }
        }
    }
remainder:
    if (ip < n) op = emit_literal(op, in + ip, n - ip);              /* snappy.cc:1039-1043 */
    return op;
}

size_t orc_snappy_compress(const uint8_t* src, size_t n, uint8_t* dst)
{
    uint8_t* op = dst;
    uint32_t v = (uint32_t)n;                                         /* Varint::Encode32 */
    while (v >= 128) { *op++ = (uint8_t)(v | 128); v >>= 7; }
    *op++ = (uint8_t)v;
    uint16_t* table = (uint16_t*)malloc(SN_MAX_TABLE * sizeof(uint16_t));
    size_t pos = 0;
    while (pos < n) {                                                 /* snappy.cc:1761-1816 */
        const uint32_t frag = (uint32_t)((n - pos < SN_BLOCK) ? n - pos : SN_BLOCK);
        op = compress_fragment(src + pos, frag, op, table, table_size_for(frag));
        pos += frag;
    }
    free(table);
    return (size_t)(op - dst);
}

int orc_snappy_uncompressed_length(const uint8_t* src, size_t n, size_t* result)
{   /* snappy.cc:1202-1223 */
    uint32_t r = 0, shift = 0;
    size_t ip = 0;
    for (;;) {
        if (shift >= 32) return 0;
        if (ip >= n) return 0;
        const uint8_t c = src[ip++];
        const uint32_t val = c & 0x7f;
        if (shift > 0 && ((val << shift) >> shift) != val) return 0;  /* LeftShiftOverflows :1144-1152 */
        r |= val << shift;
        if (c < 128) break;
        shift += 7;
    }
    *result = r;
    return (int)ip;   /* header bytes consumed (>0) */
}

int orc_snappy_uncompress(const uint8_t* src, size_t n, uint8_t* dst, size_t dstCap)
{
    size_t ulen;
    const int hdr = orc_snappy_uncompressed_length(src, n, &ulen);
    if (!hdr || ulen > dstCap) return 0;
    size_t ip = (size_t)hdr, op = 0;
    while (ip < n) {                                                  /* DecompressAllTags :1466-1570 */
        const uint8_t c = src[ip];
        const uint32_t type = c & 3;
        const uint32_t taglen = type == 0 ? ((c >> 2) < 60 ? 1 : (c >> 2) - 58) : (type == 1 ? 2 : (type == 2 ? 3 : 5));
        if (ip + taglen > n) return 0;                                /* RefillTag :1598-1653 */
        if (type == 0) {
            size_t len = (c >> 2) + 1;
            if (len > 60) {
                const uint32_t nb = (uint32_t)len - 60;
                uint32_t v = 0;
                for (uint32_t i = 0; i < nb; i++) v |= (uint32_t)src[ip + 1 + i] << (8 * i);
                len = (size_t)v + 1;
            }
            ip += taglen;
            if (len > n - ip) return 0;                               /* input exhausted inside the literal */
            if (len > ulen - op) return 0;                            /* SnappyArrayWriter::Append :2149-2158 */
            memcpy(dst + op, src + ip, len);
            ip += len; op += len;
        } else {
            size_t len, offset;
            if (type == 1) { len = 4 + ((c >> 2) & 7); offset = ((size_t)(c >> 5) << 8) | src[ip + 1]; }
            else if (type == 2) { len = (c >> 2) + 1; offset = src[ip + 1] | ((size_t)src[ip + 2] << 8); }
            else { len = (c >> 2) + 1; offset = s_rd32(src + ip + 1); }
            ip += taglen;
            if (offset == 0 || offset > op) return 0;                 /* AppendFromSelf :2189-2207 */
            if (len > ulen - op) return 0;
            for (size_t i = 0; i < len; i++) dst[op + i] = dst[op - offset + i];
            op += len;
        }
    }
    return op == ulen;                                                /* CheckLength :2146 */
}
