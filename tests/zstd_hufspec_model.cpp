// tests/zstd_hufspec_model.cpp -- TEST INFRASTRUCTURE.  Lane-by-lane CPU model of the speculative Huffman literal decoder:
// runs the kernel's own per-lane code (csrc/zstd_hufspec.h) for 16 lanes one after the other, with the repair rounds and the
// prefix sum the kernel performs between the runs, on an X1 table and a backward bitstream handed in by the test.
//
//   hufspec_model_decode(stream, n, cells, tl, nout, out, lanes, stats) -> 0 ok / -1 rejected
//     cells[1 << tl] = symbol | nbBits << 8 (HUF_DEltX1, huf_decompress.c:380-470)
//     accept  <=>  the sequential decoder reads exactly nout symbols and ends on bit 0 (what the oracle's decoder demands)
//   hufspec_seq_decode: the plain sequential decoder (one symbol at a time, bit by bit), the yardstick.
#include <stdint.h>
#include <string.h>
#include <vector>
#include "zstd_hufspec.h"

namespace {
struct Ld {
    const uint8_t* p; uint32_t n;
    uint64_t operator()(int off) const
    {
        uint64_t v = 0;
        for (int i = 0; i < 8; ++i) { const int a = off + i; if (a >= 0 && (uint32_t)a < n) v |= (uint64_t)p[a] << (8 * i); }
        return v;
    }
};
struct Tab { const uint16_t* c; uint32_t operator()(uint32_t i) const { return c[i]; } };

int top_of(const uint8_t* s, uint32_t n)
{
    if (!n || !s[n - 1]) return -1;
    int h = 7; while (!((s[n - 1] >> h) & 1)) --h;
    return (int)(n - 1) * 8 + h;
}
}

extern "C" int hufspec_seq_decode(const uint8_t* s, uint32_t n, const uint16_t* cells, uint32_t tl, uint32_t nout, uint8_t* out)
{
    int pos = top_of(s, n);
    if (pos < 0) return -1;
    auto bit = [&](int b) -> uint32_t { return (b >= 0 && b < (int)(8 * n)) ? (s[b >> 3] >> (b & 7)) & 1u : 0u; };
    for (uint32_t i = 0; i < nout; ++i) {
        uint32_t idx = 0;
        for (uint32_t k = 0; k < tl; ++k) idx = (idx << 1) | (pos - 1 - (int)k >= 0 ? bit(pos - 1 - (int)k) : 0u);
        const uint16_t c = cells[idx];
        out[i] = (uint8_t)c; pos -= c >> 8;
    }
    return pos == 0 ? 0 : -1;
}

extern "C" int hufspec_model_decode(const uint8_t* s, uint32_t n, const uint16_t* cells, uint32_t tl, uint32_t nout, uint8_t* out,
                                    int lanes, uint32_t* stats /* [0] rounds, [1] decode steps incl. the writing run */)
{
    const int top = top_of(s, n);
    if (top < 0) return -1;
    const Ld ld{s, n}; const Tab tab{cells};
    const int seg = hufspec::seg_bits(top, lanes);
    std::vector<int> start(lanes), limit(lanes), exitp(lanes);
    std::vector<uint32_t> cnt(lanes);
    std::vector<int> segTop(lanes);
    for (int j = 0; j < lanes; ++j) {                // first run: the segment's tail only (hufspec::probe_start)
        segTop[j] = hufspec::seg_start(top, seg, j); limit[j] = hufspec::seg_start(top, seg, j + 1);
        start[j] = hufspec::probe_start(segTop[j], limit[j]);
    }
    uint32_t rounds = 0, steps = 0;
    std::vector<char> dirty(lanes, 1);
    for (;;) {                                       // one round = one lock-step run of the dirty lanes
        ++rounds;
        for (int j = 0; j < lanes; ++j) if (dirty[j]) { exitp[j] = hufspec::run_count(ld, tab, tl, start[j], limit[j], &cnt[j]); steps += cnt[j]; }
        bool any = false;
        std::vector<char> nd(lanes, 0);
        for (int j = 0; j < lanes; ++j) {
            const int want = j == 0 ? segTop[0] : exitp[j - 1];
            if (want != start[j]) { start[j] = want; nd[j] = 1; any = true; }
        }
        dirty = nd;
        if (!any) break;
        if (rounds > (uint32_t)lanes + 2) return -2; // cannot happen: round r + 1 settles lane r
    }
    uint32_t total = 0;
    for (int j = 0; j < lanes; ++j) total += cnt[j];
    if (stats) { stats[0] = rounds; stats[1] = steps + total; }
    if (exitp[lanes - 1] != 0 || total != nout) return -1;
    uint32_t off = 0;
    for (int j = 0; j < lanes; ++j) {
        uint8_t* o = out + off;
        const int e = hufspec::run_write(ld, tab, tl, start[j], limit[j],
            [&](uint32_t i, uint32_t a, uint32_t b, uint32_t c, uint32_t d) { const uint32_t v[4] = {a, b, c, d}; memcpy(o + i, v, 16); },
            [&](uint32_t i, uint32_t v) { memcpy(o + i, &v, 4); }, [&](uint32_t i, uint32_t v) { o[i] = (uint8_t)v; });
        if (e != exitp[j]) return -3;
        off += cnt[j];
    }
    return 0;
}
