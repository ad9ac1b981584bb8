// csrc/zstd_hufspec.h -- per-lane pieces of the speculative Huffman literal decoder of the Zstd decoder
// (decodes what HUF_decompress4X1_usingDTable decodes, /root/reference algos/zstd/lib/decompress/huf_decompress.c:565-680,
//  bit reader lib/common/bitstream.h:262-420).
//
// A Huffman stream is a backward bitstream: `pos` unread bits, the next code is the tableLog bits right below bit `pos`.
// Sixteen lanes share one stream: the bits [0, top) are cut into 16 segments, lane j decodes the codes that START in its
// segment (start > limit_j, i.e. from its start position down to the first code start at or below the segment's lower end).
// Lane 0's start is the stream's top; lane j+1's start is lane j's exit.  Prefix codes fall into step after a few symbols
// whatever bit they start at (tools/experiments/huf_selfsync_study.py), so every lane first runs from a GUESSED start (its
// segment's upper end), then the lanes whose predecessor left somewhere else run again - until nothing changes - and a
// last run writes the symbols (tools/experiments/huf_spec_decode_model.py is the CPU model of the whole scheme).
//
// This header holds what one lane does between two exchanges - plain C++ on purpose: zstd_kernels.hip runs it per lane on the
// GPU, tests/zstd_hufspec_model.cpp runs the very same code lane after lane on the CPU against a sequential decode.
//
// The lane's view of the stream is a 128-bit register window (hi:lo = bytes [wb+8, wb+16) : [wb, wb+8)) that slides down
// 8 bytes at a time, with the next 8 bytes (pf) requested one slide ahead: the load of a slide is used ~10 symbols later.
// Bytes outside the stream read as 0, exactly as BIT_reloadDStream's tail handling leaves zeros below bit 0.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define HS_FN __device__ __forceinline__
#define HS_MEM __device__ __forceinline__
#else
#define HS_FN static inline
#define HS_MEM inline
#endif

namespace hufspec {

// Ld: uint64_t ld(int byteOffset) -> bytes [off, off+8) of the stream, little endian, 0 outside the stream.
template <class Ld>
struct Window {
    uint64_t hi, lo, pf;
    int wb;
    HS_MEM void init(Ld& ld, int pos)
    {
        wb = (pos - 64) >> 3;                       // floor: the container's low end lies in byte wb (0 <= s <= 7)
        lo = ld(wb); hi = ld(wb + 8); pf = ld(wb - 8);
    }
    // the 64 unread bits below `pos`, left-aligned (bit pos-1 at bit 63); pos may have dropped by at most 64 since the last call
    HS_MEM uint64_t container(Ld& ld, int pos)
    {
        int s = pos - 64 - 8 * wb;
        if (s < 0) { hi = lo; lo = pf; wb -= 8; pf = ld(wb - 8); s += 64; }
        return (lo >> s) | ((hi << 1) << (63 - s));
    }
};

// One run of a lane: decode from `start` while the position is above `limit`.  Tab: uint32_t tab(uint32_t index) ->
// symbol | nbBits << 8 (X1 cell); tl = table log (<= 11).  COUNT ONLY: returns the exit position, *count = symbols decoded.
template <class Ld, class Tab>
HS_FN int run_count(Ld ld, Tab tab, const uint32_t tl, const int start, const int limit, uint32_t* count)
{
    int p = start; uint32_t n = 0;
    if (p > limit) {
        Window<Ld> w; w.init(ld, p);
        const int guard = limit + 3 * (int)tl;      // above it, four codes cannot carry the position past the limit's code
        while (p > guard) {
            uint64_t c = w.container(ld, p);
#pragma unroll
            for (int k = 0; k < 4; ++k) { const uint32_t nb = tab((uint32_t)(c >> (64 - tl))) >> 8; c <<= nb; p -= (int)nb; }
            n += 4;
        }
        while (p > limit) {
            uint64_t c = w.container(ld, p);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool act = p > limit;
                const uint32_t nb = act ? tab((uint32_t)(c >> (64 - tl))) >> 8 : 0u;
                c <<= nb; p -= (int)nb; n += act ? 1u : 0u;
            }
        }
    }
    *count = n;
    return p;
}

// The writing run: same walk; St16: void st16(uint32_t index, uint32_t a, uint32_t b, uint32_t c, uint32_t d) stores 16 bytes at
// out[index..index+15] (a first), St4: void st4(uint32_t index, uint32_t fourSymbols), St1: void st1(uint32_t index, uint32_t symbol).
// Sixteen symbols per store while the segment lasts: a lane's output is a cursor of its own, and 64 cursors that advance four
// bytes at a time keep 64 cache lines half-written for long (the decoder's HBM write traffic was 3x its output).
// Returns the exit position.
template <class Ld, class Tab, class St16, class St4, class St1>
HS_FN int run_write(Ld ld, Tab tab, const uint32_t tl, const int start, const int limit, St16 st16, St4 st4, St1 st1)
{
    int p = start; uint32_t i = 0;
    if (p > limit) {
        Window<Ld> w; w.init(ld, p);
        const int guard16 = limit + 15 * (int)tl;   // above it, sixteen codes cannot carry the position past the limit's code
        while (p > guard16) {
            uint32_t acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint64_t c = w.container(ld, p);
                uint32_t a = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t cell = tab((uint32_t)(c >> (64 - tl))), nb = cell >> 8;
                    a |= (cell & 0xFFu) << (8 * k); c <<= nb; p -= (int)nb;
                }
                acc[g] = a;
            }
            st16(i, acc[0], acc[1], acc[2], acc[3]); i += 16;
        }
        const int guard = limit + 3 * (int)tl;
        while (p > guard) {
            uint64_t c = w.container(ld, p);
            uint32_t acc = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t cell = tab((uint32_t)(c >> (64 - tl))), nb = cell >> 8;
                acc |= (cell & 0xFFu) << (8 * k); c <<= nb; p -= (int)nb;
            }
            st4(i, acc); i += 4;
        }
        while (p > limit) {
            uint64_t c = w.container(ld, p);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (p > limit) {
                    const uint32_t cell = tab((uint32_t)(c >> (64 - tl))), nb = cell >> 8;
                    st1(i, cell & 0xFFu); ++i; c <<= nb; p -= (int)nb;
                }
            }
        }
    }
    return p;
}

// The FIRST run of a lane only has to find where the lane leaves its segment - the start of the lane behind it - and prefix
// codes fall into step within a few symbols: it decodes the segment's last kProbeBits bits, not the segment (round 6; the
// runs that follow, from the starts the probes found, count the symbols and confirm the exits - a probe that had not fallen
// into step yet is one more repair round, as a wrong guess always was).
#ifndef ZSTD_HUF_PROBE_BITS
#define ZSTD_HUF_PROBE_BITS 384
#endif
constexpr int kProbeBits = ZSTD_HUF_PROBE_BITS;
HS_FN int probe_start(int start, int limit) { return start - limit > kProbeBits ? limit + kProbeBits : start; }

// segment geometry: lane j of `lanes` decodes the codes that start in (limit, start]
HS_FN int seg_bits(int top, int lanes) { return (top + lanes - 1) / lanes; }
HS_FN int seg_start(int top, int seg, int j) { const int v = top - j * seg; return v > 0 ? v : 0; }

}  // namespace hufspec
