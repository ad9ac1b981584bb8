// csrc/zstd_kernels.hip -- Zstandard level 1 ("fast" strategy + Huffman/FSE entropy stage) for gfx950.
//
// One chunk (<= 128 KiB) = one single-block frame, byte-identical to the reference's
// ZSTD_compress(dst, cap, src, n, 1)  (/root/reference algos/zstd/lib/compress/zstd_compress.c:5534-5560):
//   match finder  AOCL_ZSTD_compressBlock_fast_noDict_generic   zstd_fast.c:427-706
//   entropy stage ZSTD_entropyCompressSeqStore_internal          zstd_compress.c:2910-3027
//                 (literals: zstd_compress_literals.c:129-235 + huf_compress.c; sequences:
//                  zstd_compress_sequences.c:156-382 + fse_compress.c)
//   framing       ZSTD_writeFrameHeader :4586-4632, ZSTD_compress_frameChunk :4504-4583
//
// A persistent grid of one-wave workgroups walks the chunks; each workgroup owns a slice of a global
// workspace (sequence store, literal buffer, symbol codes, staging for the compressed block).
//
// Match finder on a wavefront.  The reference probes positions in pairs (ip0, ip0+1) with a repcode test at
// ip0+step in front of every pair; the positions, the step growth (kStepIncr 128) and the order of the
// table writes depend only on the start of the search, so lane k evaluates probe k speculatively:
// hash -> LDS table -> 4-byte compare (+ repcode compare on even probes).  The batch is cut at the first
// in-batch table-slot reuse exactly as in lz4_kernels.hip, the first event in probe order wins, and only
// the table writes the reference would have executed are committed.
//
// Entropy stage.  Byte histogram (LDS atomics), then lane 0 builds the Huffman code and the three FSE
// tables with the reference's exact tie-breaking (zstd_entropy.hip.h); the Huffman streams and the
// sequence bitstream are emitted by all 64 lanes through a prefix-sum bit packer; the FSE state chains
// (inherently serial) run on three lanes, one per symbol stream, 64 sequences per tile.
#include "zstd_entropy.hip.h"
#include "host_util.h"

namespace aoclgpu {
namespace {
using namespace zs;

constexpr u32 kBlockMax = 128u << 10;
constexpr u32 kMaxSeq = kBlockMax / 4 + 8;
constexpr u32 kSchedLen = 960;

// ---- pair schedule of the fast match finder (zstd_fast.c:491-631): offset of pair m from the search
//      start, gap to the repcode probe (sigma) and the value of `step` while the pair is examined ----
struct FastSched {
    u32 a[kSchedLen]; u16 sigma[kSchedLen]; u16 st[kSchedLen];
    constexpr FastSched() : a(), sigma(), st() {
        u32 step = 2, nextStep = 128, A = 0, sig = 2;
        for (u32 m = 0; m < kSchedLen; ++m) {
            a[m] = A; sigma[m] = (u16)sig; st[m] = (u16)step;
            const u32 A2 = A + sig; sig = step;
            if (A2 + step >= nextStep) { step++; nextStep += 128; }
            A = A2;
        }
    }
};
__device__ const FastSched g_sched{};

// ---- per-workgroup workspace in global memory ----
struct Workspace {
    u8  lits[kBlockMax + 64];
    u32 seqOff[kMaxSeq];     // offBase
    u32 seqLL[kMaxSeq];      // literal length
    u32 seqML[kMaxSeq];      // match length
    u8  llc[kMaxSeq], ofc[kMaxSeq], mlc[kMaxSeq];
    u8  pad[8];
    u8  tmp[2 * kBlockMax + 1024];
};

// ---- LDS of the entropy stage (overlays the hash table once the match finder is done) ----
struct EntropyLds {
    u32 hist[256];
    u32 count[64];
    s16 norm[64];
    HNode nodes[513];
    RankPos rank[192];
    short qstack[192];
    HufCT huf;
    HufWriteScratch hw;
    FseCT ctLL, ctOF, ctML;
    u8 tableSymbol[512];
    u16 cumul[64];
    u32 tile[WavePacker::kTileWords];
    u32 fseBits[3][64];      // per tile: (value << 8) | nbBits from the three state chains
    u32 misc[16];
};

template <int MLS>
__device__ __forceinline__ u32 z_hash(u64 v, u32 hlog)
{   // ZSTD_hash5Ptr / ZSTD_hash6Ptr (zstd_compress_internal.h:843-875)
    if (MLS == 5) return (u32)(((v << 24) * 889523592379ull) >> (64 - hlog));
    return (u32)(((v << 16) * 227718039650203ull) >> (64 - hlog));
}

// common prefix of src[a..end) and src[b..], 256 bytes per step (ZSTD_count)
__device__ u32 wave_count(const u8* __restrict__ src, u32 a, u32 b, u32 end, int lane)
{
    u32 n = 0;
    for (;;) {
        const u32 o = n + 4u * (u32)lane, pa = a + o;
        u32 x = 0xFFFFFFFFu;
        if (pa + 4 <= end) x = ld32(src + pa) ^ ld32(src + b + o);
        else if (pa < end) {
            x = 0;
            for (u32 t = 0; t < end - pa; ++t) x |= (u32)(src[pa + t] ^ src[b + o + t]) << (8 * t);
            x |= 0xFFFFFFFFu << (8 * (end - pa));
        }
        const int l = first_lane(ballot(x != 0));
        if (l == 64) { n += 256; continue; }
        return n + 4u * (u32)l + ((u32)__builtin_ctz(lane_get(x, l)) >> 3);
    }
}

// ------------------------------------------------------------------------------------------------
// match finder: fills ws->seq*, ws->lits; returns nbSeq, *litSizeOut
// ------------------------------------------------------------------------------------------------
template <int MLS, typename T>
__device__ u32 fast_block(const u8* __restrict__ src, const u32 n, const u32 hlog, T* table, Workspace* ws,
                          u32* litSizeOut, int lane)
{
    const u32 tsize = 1u << hlog;
    for (u32 i = (u32)lane; i < tsize; i += kWave) table[i] = 0;
    wave_fence();
    u32 nbSeq = 0, litSize = 0, anchor = 0, s = 1, rep1 = 1, rep2 = 0;   // rep {1,4,8}: 4 > maxRep is parked (:484-489)
    if (n >= 20) {
        const u32 ilimit = n - 16;                                      // iend - HASH_READ_SIZE - PREFETCH_SAFETY (:448)
        for (;;) {                                                      // _start
            u32 i0 = 0; int ev = 0;                                     // 1 = table match, 2 = repcode
            u32 e_pos = 0, e_cand = 0, e_sig = 0, e_st = 0, e_half = 0;
            for (;;) {
                const u32 K = i0 + (u32)lane, m = K >> 1, half = K & 1;
                u32 am, sg, st;
                if (m <= 62) { am = 2 * m; sg = 2; st = 2; }
                else if (m < kSchedLen) { am = g_sched.a[m]; sg = g_sched.sigma[m]; st = g_sched.st[m]; }
                else { am = 0x40000000u; sg = 2; st = 2; }
                const bool valid = (u64)s + am + sg + 1 < ilimit;       // ip3 < ilimit (:503, :631)
                const u32 pos = s + am + half;
                u64 seq = 0;
                if (valid) seq = ld64(src + pos);
                const u32 h = z_hash<MLS>(seq, hlog);
                u32 cand = 0;
                if (valid) cand = table[h];
                bool rep_hit = false;
                if (valid && half == 0) {                               // repcode at ip2 (:519-542)
                    const u32 r = pos + sg;
                    rep_hit = ld32(src + r) == ld32(src + r - rep1);
                }
                u32 m4 = ~(u32)seq;
                if (valid && cand >= 2) m4 = ld32(src + cand - 2);
                wave_fence();
                bool pending = valid; u32 w = (u32)lane;
                do {
                    if (pending) table[h] = (T)lane;
                    wave_fence();
                    if (valid) w = table[h];
                    wave_fence();
                    pending = valid && (u32)lane < w;
                } while (ballot(pending) != 0);
                const int f = first_lane(ballot(valid && w != (u32)lane));
                const int e = first_lane(ballot(!valid));
                const int lim = f < e ? f : e;
                const bool mat_hit = m4 == (u32)seq;
                const int el = first_lane(ballot(lane < lim && (rep_hit || mat_hit)));
                if (valid) table[h] = (T)cand;
                wave_fence();
                const int ncommit = (el < 64) ? el + 1 : lim;
                if (lane < ncommit) table[h] = (T)(pos + 2);
                wave_fence();
                if (el < 64) {
                    ev = lane_get((u32)rep_hit, el) ? 2 : 1;
                    e_pos = lane_get(pos, el); e_cand = lane_get(cand, el);
                    e_sig = lane_get(sg, el); e_st = lane_get(st, el); e_half = lane_get(half, el);
                    break;
                }
                if (e < 64 && e <= f) break;
                i0 += (u32)lim;
            }
            if (!ev) break;                                             // _cleanup
            // table write of ip1 that accompanies the hit (:537, :557, :586-598)
            {
                const u32 ip1 = e_half == 0 ? e_pos + 1 : e_pos - 1 + e_sig;
                if (e_half == 0 || e_st <= 4) {
                    const u32 h1 = z_hash<MLS>(ld64(src + ip1), hlog);
                    if (lane == 0) table[h1] = (T)(ip1 + 2);
                    wave_fence();
                }
            }
            const u32 current0 = e_pos;
            u32 ip0, match0, offcode, mLength;
            if (ev == 2) {
                ip0 = e_pos + e_sig; match0 = ip0 - rep1;
                const u32 back = uni(src[ip0 - 1] == src[match0 - 1] ? 1u : 0u);
                ip0 -= back; match0 -= back; offcode = 1; mLength = 4 + back;
            } else {                                                    // _offset (:651-665)
                ip0 = e_pos; match0 = e_cand - 2;
                rep2 = rep1; rep1 = ip0 - match0; offcode = rep1 + 3; mLength = 4;
                const u32 a = ip0 - anchor, maxback = a < match0 ? a : match0;
                u32 back = 0;
                while (back < maxback) {
                    const u32 k = back + (u32)lane;
                    bool eq = false;
                    if (k < maxback) eq = src[ip0 - 1 - k] == src[match0 - 1 - k];
                    const int c = first_lane(ballot(!eq));
                    back += (u32)c;
                    if (c < 64) break;
                }
                ip0 -= back; match0 -= back; mLength += back;
            }
            mLength += wave_count(src, ip0 + mLength, match0 + mLength, n, lane);       // _match (:669)
            {
                const u32 ll = ip0 - anchor;
                if (lane == 0) { ws->seqLL[nbSeq] = ll; ws->seqOff[nbSeq] = offcode; ws->seqML[nbSeq] = mLength; }
                wave_copy(ws->lits + litSize, src + anchor, ll, lane);
                litSize += ll; nbSeq++;
            }
            ip0 += mLength; anchor = ip0;
            if (ip0 <= ilimit) {                                        // :678-702
                const u32 ha = z_hash<MLS>(ld64(src + current0 + 2), hlog);
                const u32 hb2 = z_hash<MLS>(ld64(src + ip0 - 2), hlog);
                if (lane == 0) { table[ha] = (T)(current0 + 2 + 2); table[hb2] = (T)(ip0 - 2 + 2); }
                wave_fence();
                if (rep2 > 0) {
                    while (ip0 <= ilimit && uni(ld32(src + ip0) == ld32(src + ip0 - rep2) ? 1u : 0u)) {
                        const u32 rLength = wave_count(src, ip0 + 4, ip0 + 4 - rep2, n, lane) + 4;
                        const u32 t = rep2; rep2 = rep1; rep1 = t;
                        const u32 hc = z_hash<MLS>(ld64(src + ip0), hlog);
                        if (lane == 0) {
                            table[hc] = (T)(ip0 + 2);
                            ws->seqLL[nbSeq] = 0; ws->seqOff[nbSeq] = 1; ws->seqML[nbSeq] = rLength;
                        }
                        wave_fence();
                        nbSeq++;
                        ip0 += rLength; anchor = ip0;
                    }
                }
            }
            s = ip0;
        }
    }
    wave_copy(ws->lits + litSize, src + anchor, n - anchor, lane);       // ZSTD_storeLastLiterals
    litSize += n - anchor;
    *litSizeOut = litSize;
    return nbSeq;
}

// ------------------------------------------------------------------------------------------------
// literals
// ------------------------------------------------------------------------------------------------
__device__ u32 lit_raw(u8* dst, const u8* src, u32 n, u32 type, int lane)
{   // ZSTD_noCompressLiterals / ZSTD_compressRleLiteralsBlock (zstd_compress_literals.c:39-107)
    const u32 fl = 1 + (n > 31) + (n > 4095);
    if (lane == 0) {
        if (fl == 1) dst[0] = (u8)(type + (n << 3));
        else if (fl == 2) { const u32 v = type + (1 << 2) + (n << 4); dst[0] = (u8)v; dst[1] = (u8)(v >> 8); }
        else { const u32 v = type + (3 << 2) + (n << 4); dst[0] = (u8)v; dst[1] = (u8)(v >> 8); dst[2] = (u8)(v >> 16); }
    }
    if (type == 0) { wave_copy(dst + fl, src, n, lane); return fl + n; }
    if (lane == 0) dst[fl] = src[0];
    return fl + 1;
}

// byte histogram of src[0..n) into hist[256] (LDS)
__device__ void wave_hist(u32* hist, const u8* src, u32 n, int lane)
{
    for (u32 i = (u32)lane; i < 256; i += kWave) hist[i] = 0;
    wave_fence();
    for (u32 i = (u32)lane; i < n; i += kWave) atomicAdd(&hist[src[i]], 1u);
    wave_fence();
}

// HUF_compress1X_usingCTable (huf_compress.c:1051-1116): symbols last -> first, 8 symbols per lane per round
__device__ u32 huf_encode_1x(u8* dst, const u8* src, u32 n, const HufCT* ct, u32* tile, int lane)
{
    WavePacker pk; pk.init(tile, dst, lane);
    for (u32 done = 0; done < n; done += 512) {
        // lane handles encode-order items [done + 8*lane, +8): item j is symbol src[n-1-j]
        u64 lo = 0, hi = 0; u32 nb = 0;
        const u32 j0 = done + 8u * (u32)lane;
        for (u32 k = 0; k < 8; ++k) {
            const u32 j = j0 + k;
            if (j < n) {
                const u32 sym = src[n - 1 - j];
                const u32 l = ct->nbBits[sym]; const u64 v = ct->val[sym];
                if (nb < 64) { lo |= v << nb; if (nb + l > 64) hi |= v >> (64 - nb); }
                else hi |= v << (nb - 64);
                nb += l;
            }
        }
        pk.push(lo, hi, nb, lane);
    }
    return pk.close(lane);
}
__device__ u32 huf_encode_4x(u8* dst, const u8* src, u32 n, const HufCT* ct, u32* tile, int lane)
{   // huf_compress.c:1163-1211
    const u32 seg = (n + 3) / 4;
    u32 op = 6;
    if (n < 12) return 0;
    for (int i = 0; i < 3; ++i) {
        const u32 c = huf_encode_1x(dst + op, src + (u32)i * seg, seg, ct, tile, lane);
        if (c == 0 || c > 65535) return 0;
        if (lane == 0) { dst[2 * i] = (u8)c; dst[2 * i + 1] = (u8)(c >> 8); }
        op += c;
    }
    const u32 c = huf_encode_1x(dst + op, src + 3 * seg, n - 3 * seg, ct, tile, lane);
    if (c == 0 || c > 65535) return 0;
    return op + c;
}

// HUF_compress_internal for a first block (huf_compress.c:1326-1434): 0 = not compressible, 1 = rle
__device__ u32 huf_compress(u8* dst, const u8* src, u32 n, bool single, bool suspect, EntropyLds* L, int lane)
{
    if (suspect && n >= 4096 * 10) {                                     // :1361-1375
        wave_hist(L->hist, src, 4096, lane);
        u32 lb = 0;
        for (u32 i = (u32)lane; i < 256; i += kWave) lb = lb > L->hist[i] ? lb : L->hist[i];
        for (int d = 32; d; d >>= 1) { const u32 t = __shfl_xor(lb, d); lb = lb > t ? lb : t; }
        wave_hist(L->hist, src + n - 4096, 4096, lane);
        u32 le = 0;
        for (u32 i = (u32)lane; i < 256; i += kWave) le = le > L->hist[i] ? le : L->hist[i];
        for (int d = 32; d; d >>= 1) { const u32 t = __shfl_xor(le, d); le = le > t ? le : t; }
        if (lb + le <= ((2 * 4096) >> 7) + 4) return 0;
    }
    wave_hist(L->hist, src, n, lane);
    u32 largest = 0, maxSym = 0;
    for (u32 i = (u32)lane; i < 256; i += kWave) { const u32 c = L->hist[i]; if (c > largest) largest = c; if (c) maxSym = i; }
    for (int d = 32; d; d >>= 1) {
        const u32 t = __shfl_xor(largest, d); largest = largest > t ? largest : t;
        const u32 s2 = __shfl_xor(maxSym, d); maxSym = maxSym > s2 ? maxSym : s2;
    }
    if (largest == n) { if (lane == 0) dst[0] = src[0]; return 1; }
    if (largest <= (n >> 7) + 4) return 0;
    u32 huffLog = fse_optimal_log(11, n, maxSym, 1);                     // HUF_optimalTableLog :1268-1283
    u32 h = 0;
    if (lane == 0) {
        huffLog = huf_build(&L->huf, L->hist, maxSym, huffLog, L->nodes, L->rank, L->qstack);
        h = huf_write_ctable(dst, &L->huf, maxSym, huffLog, &L->hw);
    }
    wave_fence();
    h = uni(h);
    if (!h) return 0;
    if (h + 12 >= n) return 0;
    const u32 c = single ? huf_encode_1x(dst + h, src, n, &L->huf, L->tile, lane)
                         : huf_encode_4x(dst + h, src, n, &L->huf, L->tile, lane);
    if (!c) return 0;
    if (h + c >= n - 1) return 0;                                        // HUF_compressCTable_internal :1232-1236
    return h + c;
}

// ZSTD_compressLiterals for a first block (zstd_compress_literals.c:129-235)
__device__ u32 compress_literals(u8* dst, const u8* src, u32 n, bool suspect, EntropyLds* L, int lane)
{
    const u32 lh = 3 + (n >= 1024) + (n >= 16384);
    const bool single = n < 256;
    if (n < 64) return lit_raw(dst, src, n, 0, lane);                    // ZSTD_minLiteralsToCompress = 8 << 3
    const u32 c = huf_compress(dst + lh, src, n, single, suspect, L, lane);
    if (c == 0 || c >= n - ((n >> 6) + 2)) return lit_raw(dst, src, n, 0, lane);
    if (c == 1) return lit_raw(dst, src, n, 1, lane);
    if (lane == 0) {
        if (lh == 3) { const u32 v = 2 + ((u32)(!single) << 2) + (n << 4) + (c << 14); dst[0] = (u8)v; dst[1] = (u8)(v >> 8); dst[2] = (u8)(v >> 16); }
        else if (lh == 4) { const u32 v = 2 + (2 << 2) + (n << 4) + (c << 18); st32(dst, v); }
        else { const u32 v = 2 + (3 << 2) + (n << 4) + (c << 22); st32(dst, v); dst[4] = (u8)(c >> 10); }
    }
    return lh + c;
}

// ------------------------------------------------------------------------------------------------
// sequences
// ------------------------------------------------------------------------------------------------
// ZSTD_selectEncodingType, strategy fast, first block (zstd_compress_sequences.c:156-204): 0 basic, 1 rle, 2 compressed
__device__ int select_type(u32 mostFrequent, u32 nbSeq, u32 defaultNormLog, bool defaultAllowed)
{
    if (mostFrequent == nbSeq) return (defaultAllowed && nbSeq <= 2) ? 0 : 1;
    if (defaultAllowed) {
        const u32 dynMin = ((1u << defaultNormLog) * 9) >> 3;
        if (nbSeq < dynMin || mostFrequent < (nbSeq >> (defaultNormLog - 1))) return 0;
    }
    return 2;
}

// histogram + mode + table description + CTable of one symbol stream (ZSTD_buildCTable :242-288).
// All lanes call it; returns type (or -1), advances *op, updates *lastCount.
__device__ int build_stream(u8* base, u32* op, FseCT* ct, const u8* codes, u32 nbSeq, u32 maxCode, u32 fseLog,
                            const s16* defNorm, u32 defLog, u32 defMax, u32* lastCount, EntropyLds* L, int lane)
{
    for (u32 i = (u32)lane; i < 64; i += kWave) L->count[i] = 0;
    wave_fence();
    for (u32 i = (u32)lane; i < nbSeq; i += kWave) atomicAdd(&L->count[codes[i]], 1u);
    wave_fence();
    int type = 0; u32 adv = 0, lc = 0;
    if (lane == 0) {
        u32 max = 0, most = 0;
        for (u32 i = 0; i <= maxCode; i++) { if (L->count[i]) max = i; if (L->count[i] > most) most = L->count[i]; }
        const bool defaultAllowed = (defMax == 28) ? (max <= 28) : true;              // zstd_compress.c:2847-2848
        type = select_type(most, nbSeq, defLog, defaultAllowed);
        if (type == 1) { fse_build_rle(ct, max); base[*op] = codes[0]; adv = 1; }
        else if (type == 0) {
            for (u32 i = 0; i <= defMax; i++) L->norm[i] = defNorm[i];
            fse_build_ctable(ct, L->norm, defMax, defLog, L->tableSymbol, L->cumul);
        } else {
            u32 nbSeq1 = nbSeq;
            const u32 tableLog = fse_optimal_log(fseLog, nbSeq, max, 2);
            const u32 last = codes[nbSeq - 1];
            if (L->count[last] > 1) { L->count[last]--; nbSeq1--; }
            if (fse_normalize(L->norm, tableLog, L->count, nbSeq1, max, nbSeq1 >= 2048) < 0) type = -1;
            else {
                const u32 h = fse_write_ncount(base + *op, L->norm, max, tableLog);
                if (!h) type = -1;
                else { fse_build_ctable(ct, L->norm, max, tableLog, L->tableSymbol, L->cumul); adv = h; lc = h; }
            }
        }
    }
    wave_fence();
    type = (int)uni((u32)type); adv = uni(adv); lc = uni(lc);
    *op += adv;
    if (lc) *lastCount = lc;
    return type;
}

// ZSTD_entropyCompressSeqStore_internal (zstd_compress.c:2910-3027): block content size, 0 = emit raw
__device__ u32 entropy_compress(u8* dst, Workspace* ws, u32 litSize, u32 nbSeq, EntropyLds* L, int lane)
{
    u32 op = 0;
    const bool suspect = (nbSeq == 0) || (litSize / nbSeq >= 20);
    op += compress_literals(dst, ws->lits, litSize, suspect, L, lane);
    if (lane == 0) {
        if (nbSeq < 128) dst[op] = (u8)nbSeq;
        else if (nbSeq < 0x7F00) { dst[op] = (u8)((nbSeq >> 8) + 0x80); dst[op + 1] = (u8)nbSeq; }
        else { dst[op] = 0xFF; dst[op + 1] = (u8)(nbSeq - 0x7F00); dst[op + 2] = (u8)((nbSeq - 0x7F00) >> 8); }
    }
    op += nbSeq < 128 ? 1 : (nbSeq < 0x7F00 ? 2 : 3);
    if (nbSeq == 0) return op;
    for (u32 i = (u32)lane; i < nbSeq; i += kWave) {                     // ZSTD_seqToCodes :2715-2741
        const u32 ll = ws->seqLL[i], ml = ws->seqML[i] - 3;
        ws->llc[i] = (u8)(ll > 0xFFFF ? 35 : ll_code(ll));
        ws->ofc[i] = (u8)hb(ws->seqOff[i]);
        ws->mlc[i] = (u8)(ml > 0xFFFF ? 52 : ml_code(ml));
    }
    wave_fence();
    const u32 seqHead = op++;
    u32 lastCount = 0;
    const int tLL = build_stream(dst, &op, &L->ctLL, ws->llc, nbSeq, 35, 9, kLLnorm, 6, 35, &lastCount, L, lane);
    if (tLL < 0) return 0;
    const int tOF = build_stream(dst, &op, &L->ctOF, ws->ofc, nbSeq, 31, 8, kOFnorm, 5, 28, &lastCount, L, lane);
    if (tOF < 0) return 0;
    const int tML = build_stream(dst, &op, &L->ctML, ws->mlc, nbSeq, 52, 9, kMLnorm, 6, 52, &lastCount, L, lane);
    if (tML < 0) return 0;
    if (lane == 0) dst[seqHead] = (u8)((tLL << 6) + (tOF << 4) + (tML << 2));

    // ZSTD_encodeSequences_body (zstd_compress_sequences.c:290-382).  Encode order e = 0..nbSeq-1 handles
    // sequence i = nbSeq-1-e.  Lanes 0,1,2 run the LL / ML / OF state chains for 64 sequences, then all lanes
    // pack {OF state bits, ML state bits, LL state bits, LL extra, ML extra, OF extra} of one sequence each.
    WavePacker pk; pk.init(L->tile, dst + op, lane);
    const FseCT* myct = lane == 0 ? &L->ctLL : (lane == 1 ? &L->ctML : &L->ctOF);
    const u8* mycodes = lane == 0 ? ws->llc : (lane == 1 ? ws->mlc : ws->ofc);
    u32 state = 0;
    if (lane < 3) state = fse_init_state(myct, mycodes[nbSeq - 1]);
    for (u32 e0 = 0; e0 < nbSeq; e0 += kWave) {
        const u32 cnt = nbSeq - e0 < kWave ? nbSeq - e0 : kWave;
        if (lane < 3) {
            for (u32 k = 0; k < cnt; ++k) {
                const u32 e = e0 + k;
                u32 out = 0;
                if (e > 0) {
                    const u32 sym = mycodes[nbSeq - 1 - e];
                    const u32 nb = (state + (u32)myct->deltaNbBits[sym]) >> 16;
                    out = ((state & ((1u << nb) - 1)) << 8) | nb;
                    state = myct->stateTable[(int)(state >> nb) + myct->deltaFindState[sym]];
                }
                L->fseBits[lane][k] = out;
            }
        }
        wave_fence();
        u64 lo = 0, hi = 0; u32 nb = 0;
        if ((u32)lane < cnt) {
            const u32 i = nbSeq - 1 - (e0 + (u32)lane);
            const u32 fl = L->fseBits[0][lane], fm = L->fseBits[1][lane], fo = L->fseBits[2][lane];
            const u32 llb = kLLbits[ws->llc[i]], mlb = kMLbits[ws->mlc[i]], ofb = ws->ofc[i];
            auto put = [&](u64 v, u32 l) {
                if (!l) return;
                v &= (1ull << l) - 1;
                if (nb < 64) { lo |= v << nb; if (nb + l > 64) hi |= v >> (64 - nb); }
                else hi |= v << (nb - 64);
                nb += l;
            };
            put(fo >> 8, fo & 0xFF); put(fm >> 8, fm & 0xFF); put(fl >> 8, fl & 0xFF);
            put(ws->seqLL[i], llb); put(ws->seqML[i] - 3, mlb); put(ws->seqOff[i], ofb);
        }
        pk.push(lo, hi, nb, lane);
        wave_fence();
    }
    {   // FSE_flushCState x3: ML, OF, LL (:371-373)
        const u32 sLL = lane_get(state, 0), sML = lane_get(state, 1), sOF = lane_get(state, 2);
        u64 v = 0; u32 nb = 0;
        if (lane == 0) {
            const u32 lm = (u32)L->ctML.tableLog, lo_ = (u32)L->ctOF.tableLog, ll_ = (u32)L->ctLL.tableLog;
            v = (u64)(sML & ((1u << lm) - 1));
            v |= (u64)(sOF & ((1u << lo_) - 1)) << lm;
            v |= (u64)(sLL & ((1u << ll_) - 1)) << (lm + lo_);
            nb = lm + lo_ + ll_;
        }
        pk.push(v, 0, nb, lane);
    }
    const u32 bs = pk.close(lane);
    op += bs;
    if (lastCount && lastCount + bs < 4) return 0;                       // zstd_compress.c:3017-3023
    return op;
}

// cParams for level 1 and a known size (clevels.h:80,107 + ZSTD_adjustCParams_internal zstd_compress.c:1575-1595)
__device__ __forceinline__ void level1_params(u32 n, u32* hlog, u32* mls)
{
    const u32 srcLog = (n < 64) ? 6 : (u32)hb(n - 1) + 1;
    if (n <= 16384) { const u32 w = srcLog < 14 ? srcLog : 14; *hlog = (15 < w + 1) ? 15 : w + 1; *mls = 5; }
    else { const u32 w = srcLog < 17 ? srcLog : 17; *hlog = (13 < w + 1) ? 13 : w + 1; *mls = 6; }
}

__device__ int zstd_compress_wave(const u8* __restrict__ src, const u32 n, u8* __restrict__ dst, const u32 cap,
                                  void* lds, u32 ldsBytes, Workspace* ws, int lane)
{
    const u32 bound = n + (n >> 8) + (n < kBlockMax ? ((kBlockMax - n) >> 11) : 0);
    if (n > kBlockMax || cap < bound || cap < 9) return 0;
    u32 op = 0;
    const u32 fcs = (n >= 256) + (n >= 65536 + 256);
    if (lane == 0) {                                                     // ZSTD_writeFrameHeader :4586-4632
        dst[0] = 0x28; dst[1] = 0xB5; dst[2] = 0x2F; dst[3] = 0xFD;
        dst[4] = (u8)((1 << 5) | (fcs << 6));
        if (fcs == 0) dst[5] = (u8)n;
        else if (fcs == 1) { const u32 v = n - 256; dst[5] = (u8)v; dst[6] = (u8)(v >> 8); }
        else st32(dst + 5, n);
    }
    op = 5 + (fcs == 0 ? 1 : (fcs == 1 ? 2 : 4));
    if (n == 0) { if (lane == 0) { dst[op] = 1; dst[op + 1] = 0; dst[op + 2] = 0; } return (int)(op + 3); }   // ZSTD_writeEpilogue
    u32 cSize = 0;
    if (n >= 7) {                                                        // ZSTD_buildSeqStore :3268
        u32 hlog, mls, litSize = 0, nbSeq;
        level1_params(n, &hlog, &mls);
        const u32 need = (mls == 5 ? 2u : 4u) << hlog;
        if (need > ldsBytes) return 0;                                   // launcher sizes LDS per chunk class
        if (mls == 5) nbSeq = fast_block<5, u16>(src, n, hlog, reinterpret_cast<u16*>(lds), ws, &litSize, lane);
        else nbSeq = fast_block<6, u32>(src, n, hlog, reinterpret_cast<u32*>(lds), ws, &litSize, lane);
        wave_fence();
        cSize = entropy_compress(ws->tmp, ws, litSize, nbSeq, reinterpret_cast<EntropyLds*>(lds), lane);
        if (cSize >= n - ((n >> 6) + 2)) cSize = 0;                      // ZSTD_minGain, zstd_compress.c:3055-3057
    }
    if (cSize == 0) {                                                    // ZSTD_noCompressBlock
        const u32 h = 1 + (0 << 1) + (n << 3);
        if (lane == 0) { dst[op] = (u8)h; dst[op + 1] = (u8)(h >> 8); dst[op + 2] = (u8)(h >> 16); }
        wave_copy(dst + op + 3, src, n, lane);
        return (int)(op + 3 + n);
    }
    const u32 h = 1 + (2 << 1) + (cSize << 3);
    if (lane == 0) { dst[op] = (u8)h; dst[op + 1] = (u8)(h >> 8); dst[op + 2] = (u8)(h >> 16); }
    wave_fence();
    wave_copy(dst + op + 3, ws->tmp, cSize, lane);
    return (int)(op + 3 + cSize);
}

}  // namespace

__global__ void __launch_bounds__(64) zstd_compress_kernel(aoclgpu_batch b, uint8_t* wsBase, uint32_t ldsBytes,
                                                           uint32_t first, uint32_t count)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    Workspace* ws = reinterpret_cast<Workspace*>(wsBase) + blockIdx.x;
    const int lane = lane_id();
    for (uint32_t c = first + blockIdx.x; c < first + count; c += gridDim.x) {
        const ChunkView v = compress_view(b, c);
        const int r = zstd_compress_wave(v.src, v.src_len, v.dst, v.dst_cap, lds, ldsBytes, ws, lane);
        if (lane == 0) b.out_len[c] = r;
        wave_fence();
    }
}

size_t zstd_workspace_bytes() { return sizeof(Workspace); }
size_t zstd_entropy_lds_bytes() { return sizeof(EntropyLds); }

}  // namespace aoclgpu

// ------------------------------------------------------------------------------------------------
// C-ABI launchers
// ------------------------------------------------------------------------------------------------
namespace {
struct WsPool { void* p = nullptr; size_t bytes = 0; int dev = -1; };
WsPool g_pool;

void* get_workspace(size_t bytes)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (g_pool.p && g_pool.dev == dev && g_pool.bytes >= bytes) return g_pool.p;
    if (g_pool.p) { (void)hipFree(g_pool.p); g_pool.p = nullptr; }
    if (hipMalloc(&g_pool.p, bytes) != hipSuccess) { g_pool.p = nullptr; return nullptr; }
    g_pool.bytes = bytes; g_pool.dev = dev;
    return g_pool.p;
}
uint32_t lds_for_chunk(uint64_t n)
{   // hash table bytes of the level-1 parameters for an n-byte chunk, never less than the entropy-stage LDS
    uint32_t need;
    if (n <= 16384) {
        uint32_t srcLog = 6; if (n >= 64) { srcLog = 0; uint64_t t = n - 1; while (t) { srcLog++; t >>= 1; } }
        const uint32_t w = srcLog < 14 ? srcLog : 14, hl = (15 < w + 1) ? 15 : w + 1;
        need = 2u << hl;
    } else need = 4u << 13;
    const uint32_t ent = (uint32_t)aoclgpu::zstd_entropy_lds_bytes();
    need = need > ent ? need : ent;
    return (need + 255u) & ~255u;
}
}  // namespace

extern "C" int aoclgpu_zstd_compress_chunks(const aoclgpu_batch* b, void* stream)
{
    if (int rc = aoclgpu::check_batch(b, false)) return rc;
    if (b->nchunks == 0) return AOCLGPU_OK;
    hipStream_t s = (hipStream_t)stream;
    // launch classes: uniform mode -> body chunks share one LDS size, the ragged tail chunk gets its own launch;
    // explicit mode -> sizes live on the device, so every block gets the largest table (64 KiB)
    struct Launch { uint32_t first, count, lds; } L[2]; int nl = 0;
    if (b->src_off) { L[nl++] = { 0, b->nchunks, 65536u }; }
    else {
        const uint64_t tail = b->total_bytes - (uint64_t)(b->nchunks - 1) * b->chunk_bytes;
        if (b->chunk_bytes > (128u << 10) && b->nchunks > 1) { aoclgpu::set_error_msg("zstd: chunk_bytes > 128 KiB"); return AOCLGPU_ERR_ARG; }
        if (b->nchunks > 1) L[nl++] = { 0, b->nchunks - 1, lds_for_chunk(b->chunk_bytes) };
        L[nl++] = { b->nchunks - 1, 1, lds_for_chunk(tail) };
    }
    for (int i = 0; i < nl; ++i) {
        const uint32_t perCU = 160u * 1024u / L[i].lds;
        uint32_t grid = 256u * (perCU ? perCU : 1u);
        if (grid > L[i].count) grid = L[i].count;
        if (grid > 2048) grid = 2048;
        void* ws = get_workspace((size_t)grid * aoclgpu::zstd_workspace_bytes());
        if (!ws) { aoclgpu::set_error_msg("zstd: workspace allocation failed"); return AOCLGPU_ERR_HIP; }
        if (L[i].lds > 65536u) {
            hipError_t e = hipFuncSetAttribute((const void*)aoclgpu::zstd_compress_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L[i].lds);
            if (e != hipSuccess) { aoclgpu::set_error("hipFuncSetAttribute", e); return AOCLGPU_ERR_HIP; }
        }
        hipLaunchKernelGGL(aoclgpu::zstd_compress_kernel, dim3(grid), dim3(64), L[i].lds, s, *b, (uint8_t*)ws, L[i].lds, L[i].first, L[i].count);
        if (int rc = aoclgpu::check_launch("zstd_compress_kernel")) return rc;
    }
    return AOCLGPU_OK;
}

extern "C" int aoclgpu_zstd_decompress_chunks(const aoclgpu_batch*, void*)
{
    aoclgpu::set_error_msg("zstd decompress: not implemented");
    return AOCLGPU_ERR_ARG;
}
