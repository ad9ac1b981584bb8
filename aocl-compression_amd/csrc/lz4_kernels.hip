// csrc/lz4_kernels.hip -- LZ4 block format, level 1, for gfx950 (MI355X).
//
// Bit-exact wave-parallel restatement of the reference's greedy parser and a safe decoder:
//   compress   : AOCL_LZ4_compress_generic_validated   /root/reference algos/lz4/lz4.c:1371-1845
//                (entry LZ4_compress_default :2967-2975; table-type / limitedOutput selection :2513-2541)
//   decompress : AOCL_LZ4_decompress_generic (safe)    algos/lz4/lz4.c:3367-3803, entry :4898-4923
//
// How a sequential greedy parse runs on a 64-lane wavefront without changing one output byte
// ------------------------------------------------------------------------------------------
// Between two matches the reference visits a data-INDEPENDENT schedule of probe positions
// (step = searchMatchNb++ >> 6, lz4.c:1490-1510).  Lane k of the wave therefore evaluates probe k of
// the schedule speculatively: hash, LDS table lookup, 4-byte compare.  The only data dependence
// inside a batch is through the hash table (probe k must see the table writes of probes < k), so
//   1. the first lane f whose hash slot was already written by an earlier lane of the same batch is
//      found exactly (per-slot minimum via LDS write/read-back rounds);
//   2. lanes < f have exactly the candidates the sequential code would see; the first of them with a
//      valid 4-byte match (ballot + ctz) is the reference's match;
//   3. only the table writes of probes the reference would have executed are committed
//      (lanes <= match lane, or lanes < f when the batch is cut), then the next batch starts at f.
// The "test next position" probe after a match (lz4.c:1763-1800) is probe -1 of the next batch, and
// the ip-2 back-fill (:1750) is applied before that batch reads the table.  Backward catch-up and
// forward match extension compare 64 / 256 bytes per step with ballot; token, length bytes and
// literals are written cooperatively.
#include "common.hip.h"
#include "lz_group.hip.h"
#include "host_util.h"

namespace aoclgpu {
namespace {

constexpr uint32_t kMinMatch = 4, kLastLiterals = 5, kMfLimit = 12, kMinLength = 13;
constexpr uint32_t kDistMax = 65535, k64KLimit = 65536 + 11, kMaxInput = 0x7E000000u;
constexpr uint32_t kTableBytes = 16384;   // LZ4_MEMORY_USAGE 14 (lz4.h:163)
constexpr uint32_t kSrcRing = 544;        // compressor's source ring: 2 x 256 B + 32 B mirror
// Sequence records: the parser does not write the block itself.  Every sequence it decides becomes a record (literal
// length, match length - 4, offset) in a small LDS buffer behind the source ring - the bytes that were left of the chunk's
// LDS granules - and when the buffer may overflow in the next batch the wave formats all buffered sequences at once, ONE
// SEQUENCE PER LANE (prefix sums give every sequence its literal source and its output position).  A batch adds at most 16
// sequences (64 consecutive positions, a sequence covers at least 4) plus one from the one-at-a-time path.
constexpr uint32_t kRecCap = 60, kRecFlush = kRecCap - 17;
constexpr uint32_t rec_bytes(int mode) { return mode == 1 ? 8u : 16u; }   // byU32 tables above 256 KiB: 32-bit fields
#ifndef LZ4_LDS_PAD
#define LZ4_LDS_PAD 0
#endif

__device__ __forceinline__ uint32_t lz4_bound(uint32_t n) { return n + n / 255 + 16; }   // lz4.h:261

// Sum of the first i steps of the search schedule (lz4.c:1486-1510): step_0 = 1, step_j = (64*a - 1 + j) >> 6 for
// acceleration a (LZ4_compress_fast; a = 1 for LZ4_compress_default).  With S(x) = sum of (m >> 6) over m < x this is
// 1 + S(64a + i - 1) - S(64a); saturates at 2^32 - 1 (positions that large are beyond any chunk).
__device__ __forceinline__ uint32_t sched(uint32_t i, uint32_t a)
{
    if (i == 0) return 0;
    if (a == 1) {
        const uint32_t n = 62 + i, q = n >> 6, r = n & 63;
        return 1 + 32 * q * (q - 1) + q * (r + 1);
    }
    const uint64_t x = 64ull * a + i - 1, q = x >> 6, r = x & 63;
    const uint64_t v = 1 + 32 * q * (q - 1) + q * r - 32ull * a * (a - 1);
    return v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)v;
}

template <bool U16>
__device__ __forceinline__ uint32_t lz4_hash(uint64_t seq)
{   // lz4.c:759-823: byU16 -> 13-bit hash of 4 bytes; byU32 -> 12-bit hash of the low 5 bytes
    if (U16) return ((uint32_t)seq * 2654435761u) >> (32 - 13);
    return (uint32_t)(((seq << 24) * 889523592379ull) >> (64 - 12));
}

// the same hash of the 8 bytes lo | hi << 32 in 32-bit arithmetic: (seq << 24) * P = seq * (P << 24) mod 2^64, and only the
// product's upper half is looked at
template <bool U16>
__device__ __forceinline__ uint32_t lz4_hash2(uint32_t lo, uint32_t hi)
{
    if (U16) return (lo * 2654435761u) >> (32 - 13);
    constexpr uint64_t K = 889523592379ull << 24;
    constexpr uint32_t Klo = (uint32_t)K, Khi = (uint32_t)(K >> 32);
    return (__umulhi(lo, Klo) + lo * Khi + hi * Klo) >> 20;
}

// write the LZ4 "length - 15" continuation bytes (255,255,...,rem); returns bytes written
__device__ __forceinline__ uint32_t put_len_bytes(gbyte* d, uint32_t rem, int lane)
{
    const uint32_t nff = rem / 255;
    for (uint32_t i = (uint32_t)lane; i < nff; i += kWave) d[i] = 255;
    if (lane == 0) d[nff] = (uint8_t)(rem - nff * 255);
    return nff + 1;
}

#ifdef AOCLGPU_EXPERIMENTS
__device__ unsigned long long g_lz4_prof2[8];     // -DLZ4_PROF_SPLIT: stage 4 in pieces (walk, anchors + cover, commit = what is left in stage 4)
__device__ unsigned long long g_lz4_prof[16];     // per-stage cycle sums of the PROF instantiation (tools/lz4_stage_prof.py)
__device__ unsigned long long g_lz4_cond[8];      // conditional lanes: could-match duplicates, conditional ones, stops taken, stops cut, cut with no written lane on the slot
#define LZ4_CC(k, n) (ccnt[k] += (n))
#else
#define LZ4_CC(k, n) ((void)0)
#endif

// 16-byte window [p-4, p+12) as two 64-bit halves (bytes before the chunk start read as 0)
struct Win { uint64_t lo, hi; };
__device__ __forceinline__ Win load_win(const gbyte* __restrict__ src, uint32_t p)
{   // ONE 16-byte load per lane: scattered loads cost the address unit a cycle per lane and instruction, not per byte
    typedef u32x4 __attribute__((aligned(1))) U;
    const uint32_t q = p >= 4 ? p - 4 : 0;
    const u32x4 v = *(const __attribute__((address_space(1))) U*)(src + q);
    Win w;
    w.lo = v.x | ((uint64_t)v.y << 32); w.hi = v.z | ((uint64_t)v.w << 32);
    if (p < 4) {                                          // bytes before the chunk start read as 0
        const uint32_t sh = 8 * (4 - p);
        w.hi = (w.hi << sh) | (w.lo >> (64 - sh));
        w.lo <<= sh;
    }
    return w;
}

// MODE 0: byU16 table (chunks < 65547 B), u16[8192].  MODE 1: byU32 table packed as u16[4096] + 2 extra bits per
// slot (chunks <= 256 KiB): 9 KB of LDS instead of 16 KB, so 17 instead of 10 chunks are resident per CU.
// MODE 2: byU32 table as u32[4096] whose spare high bits hold check bits (any chunk size).
//
// LINKED (MODE 2 only): one block [lk->b0, n) of a STREAM whose bytes start at src[0] (LZ4_compress_fast_continue, lz4.c:3179-3254,
// as LZ4F_compressUpdate / a frame with linked blocks drives it).  The table persists between blocks and holds stream indices
// (window position + idxBase, no check bits), candidates below lowIdx are outside the dictionary (dictSmall, :1000, :1137) and
// further than 64 KB are too far (:1124-1127), the catch-up stops at lowPos - the start of the dictionary - or, with ext
// (usingExtDict: the block does not follow its dictionary in the caller's memory), at the block's start for a match inside the
// block (:1112-1119, :1149).  Everything else - batches, records, formatter - is the chunk kernels' machinery.
struct Lz4Link { uint32_t b0, idxBase, lowIdx, lowPos, ext; };
template <int MODE, bool PROF, bool LINKED = false>
__device__ __forceinline__ int lz4_compress_wave(const gbyte* __restrict__ src_, const uint32_t n_,
                                 gbyte* __restrict__ dst_, const uint32_t cap_, lbyte* lds_, const uint32_t expf_,
                                 uint32_t* last_run, const Lz4Link* lk = nullptr)
{
    static_assert(!LINKED || MODE == 2, "linked blocks use the u32 table");
    const uint32_t b0 = LINKED ? uni(lk->b0) : 0u, idxBase = LINKED ? uni(lk->idxBase) : 0u, lowIdx = LINKED ? uni(lk->lowIdx) : 0u;
    const uint32_t lowPos = LINKED ? uni(lk->lowPos) : 0u; const bool ext = LINKED && uni(lk->ext) != 0;
    const gbyte* __restrict__ src = uni_ptr(src_);
    gbyte* __restrict__ dst = uni_ptr(dst_);
    lbyte* lds = uni_ptr(lds_);
    const uint32_t n = uni(n_), cap = uni(cap_), expf = uni(expf_);
    const uint32_t accel = (expf >> 15) + 1;              // AOCLGPU_LZ4_ACCELERATION(a) in batch.reserved
    constexpr bool U16 = (MODE == 0);
    using T = typename std::conditional<MODE == 2, uint32_t, uint16_t>::type;
    typedef __attribute__((address_space(3))) T LT;
    typedef __attribute__((address_space(3))) uint32_t L32;
    LT* table = (LT*)lds;
    L32* hib = (L32*)(lds + 8192);                        // MODE 1: 256 words
    auto tget = [&](uint32_t h) -> uint32_t {
        uint32_t v = table[h];
        if (MODE == 1) v |= ((hib[h >> 4] >> ((h & 15) * 2)) & 3u) << 16;
        return v;
    };
    auto tput = [&](uint32_t h, uint32_t v, uint32_t old) {           // `old` = entry this lane read from the slot
        table[h] = (T)v;
        if (MODE == 1) { const uint32_t x = ((old ^ v) >> 16) & 3u; if (x) __hip_atomic_fetch_xor(&hib[h >> 4], x << ((h & 15) * 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    };
    auto tput_blind = [&](uint32_t h, uint32_t v) {
        table[h] = (T)v;
        if (MODE == 1) {                                              // one masked-or instead of an AND and an OR: word = (word & ~mask) | bits
            const uint32_t sh = (h & 15) * 2;
            asm volatile("ds_mskor_b32 %0, %1, %2" :: "v"((uint32_t)(uintptr_t)&hib[h >> 4]), "v"(3u << sh), "v"(((v >> 16) & 3u) << sh) : "memory");
        }
    };
    const int lane = lane_id();
#ifdef AOCLGPU_EXPERIMENTS
    uint32_t ccnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    uint64_t tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // PROF: batches, sequences, stop: beyond-batch, lim, more_fwd, nomatch, sum(lim), -
    if (PROF) tprev = __builtin_amdgcn_s_memtime();
#ifdef LZ4_PROF_NOWAIT   /* stamps without draining the global loads: a stage is charged the waits the compiler placed in it */
#define STAGE(k) do { if (PROF) { const uint64_t t_ = __builtin_amdgcn_s_memtime(); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tacc[k] += t_ - tprev; tprev = t_; } } while (0)
#else
#define STAGE(k) do { if (PROF) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const uint64_t t_ = __builtin_amdgcn_s_memtime(); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tacc[k] += t_ - tprev; tprev = t_; } } while (0)
#endif

#if defined(AOCLGPU_EXPERIMENTS) && defined(LZ4_PROF_SPLIT)
    uint64_t tacc2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAGE2(k) do { if (PROF) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const uint64_t t_ = __builtin_amdgcn_s_memtime(); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tacc2[k] += t_ - tprev; tprev = t_; } } while (0)
#else
#define STAGE2(k) do { } while (0)
#endif

    // Table entries of the u32 table carry, above the position, a few "check" bits derived from the 4 bytes at
    // that position.  They never change a decision (a probe whose check bits differ cannot pass the 4-byte test);
    // they only spare the 64 speculative lanes a cache-line fetch each for candidates that cannot match.
    // posBits = bits needed for a position of this chunk; an empty slot aliases position 0 exactly as in the
    // reference's zero-filled table (lz4.c:2519), so the table is filled with entry(0).
    // MODE 1 (round 5): 17 position bits + ONE CHECK BIT in the 18 bits of an entry.  A chunk of up to 256 KiB needs 18 bits for a
    // position, but a candidate further than 65535 bytes back is rejected anyway (lz4.c:1556-1564): an entry is kept modulo 2^17
    // and stands for the latest position with that residue - exact as long as no entry is older than 2^17 when it is read, which
    // the age sweeps below see to (a chunk of at most 128 KiB never needs one).  The freed bit holds a hash bit of the 4 bytes at
    // the entry's position: half of the candidates that cannot match are not fetched at all (-45 % FETCH_SIZE, -6 % time on 128
    // KiB chunks where the bit was free to try: profiles/r05_kbench_steps.txt).  -DLZ4_NO_CHK1 restores the plain 18-bit entries.
#ifdef LZ4_NO_CHK1
    constexpr bool kChk1 = false;
#else
    constexpr bool kChk1 = MODE == 1;
#endif
    const uint32_t posBits = LINKED ? 32u : (MODE == 0 ? 16u : (MODE == 1 ? (kChk1 ? 17u : 18u) : (32u - (uint32_t)__builtin_clz(n - 1))));
    const uint32_t posMask = (posBits >= 32) ? 0xFFFFFFFFu : ((1u << posBits) - 1);
    auto chk = [&](uint32_t v4) -> uint32_t { return kChk1 ? (((v4 * 2654435761u) >> 31) << 17) : ((MODE != 2 || LINKED) ? 0u : (((v4 * 2654435761u) >> posBits) << posBits)); };
    // table entry <-> position: a chunk's entry is its position (| check bits), a stream's the position's stream index
    auto entry_of = [&](uint32_t p, uint32_t c) -> uint32_t { return LINKED ? p + idxBase : ((kChk1 ? (p & 0x1FFFFu) : p) | c); };
    // kChk1: the candidate of a probe at p is the latest position below p with the entry's residue (distance 1 .. 2^17)
    auto cand_of = [&](uint32_t e, uint32_t p) -> uint32_t {
        if (kChk1) { const uint32_t d = (p - e) & 0x1FFFFu; return p - (d ? d : 0x20000u); }
        return LINKED ? e - idxBase : (e & posMask);
    };
    auto usable = [&](uint32_t e, uint32_t c, uint32_t p, uint32_t mychk) -> bool {   // can this entry still pass the 4-byte test at p
        if (LINKED) return e >= lowIdx && c + kDistMax >= p;
        if (kChk1) return ((e ^ mychk) & 0x20000u) == 0 && p - c <= kDistMax;          // (p - c = 1 .. 2^17)
        return (MODE != 2 || ((e ^ mychk) & ~posMask) == 0) && (U16 || c + kDistMax >= p);
    };
    // Age sweeps (kChk1, chunks above 128 KiB).  age_sweep(ref): every entry more than 65535 bytes behind `ref` - dead for every
    // probe at or behind ref - is replaced by one exactly 65536 behind it.  The main loop sweeps at its lowest read position
    // whenever that has moved 32 KiB since the last sweep, so an entry is at most 65536 + 32 KiB + a batch's reach (< 8 KB) old
    // when it is read and at most 65536 + 40 KiB when the next sweep looks at it: never 2^17, where it would look young again.
    // Entries written by the lanes of the previous batch may lie a few bytes AHEAD of ref: their distance comes out just below
    // 2^17 (or 0) and is taken for what it is.  The one step that moves far in one go - a long match of the one-at-a-time path -
    // sweeps at its own position for what will be dead at its END (jump), or wipes the table when the match alone is longer than
    // any candidate may be away (kill_all).  A lane takes the 16 slots of one word of extra bits; ~8 sweeps per 256 KiB chunk,
    // 0.2 % of its time.  (Found by the campaign, seed 19130: a match of 63 470 bytes whose successor matches 65 535 bytes back.)
    uint32_t sweep_ref = b0;
    constexpr uint32_t kSweepEvery = 32768u, kSweepAhead = 256u, kJumpMax = 60000u;
    auto age_sweep = [&](uint32_t ref, uint32_t jump) {                // every entry is at most 2^17 - kSweepAhead behind ref, none more than kSweepAhead ahead of it
        typedef __attribute__((address_space(3))) u32x4 LQ;              // afterwards: nothing is more than 65536 behind ref + jump
        const uint32_t mk = (ref + jump - 65536u) & 0x1FFFFu;
        for (uint32_t w = (uint32_t)lane; w < 256; w += kWave) {
            uint32_t hb = hib[w];
            u32x4 t0 = *(LQ*)(lds + 32u * w), t1 = *(LQ*)(lds + 32u * w + 16);
            uint32_t tv[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t low = (tv[k >> 1] >> (16 * (k & 1))) & 0xFFFFu, c17 = low | (((hb >> (2 * k)) & 1u) << 16);
                const uint32_t d = (ref - c17) & 0x1FFFFu;
                const int32_t age = d >= 0x20000u - kSweepAhead ? (int32_t)d - 0x20000 : (int32_t)d;   // (negative: written by the batch before, ahead of ref)
                if (age + (int32_t)jump > (int32_t)kDistMax) {
                    tv[k >> 1] = (k & 1) ? ((tv[k >> 1] & 0xFFFFu) | (mk << 16)) : ((tv[k >> 1] & 0xFFFF0000u) | (mk & 0xFFFFu));
                    hb = (hb & ~(3u << (2 * k))) | ((mk >> 16) << (2 * k));
                }
            }
            *(LQ*)(lds + 32u * w) = u32x4{tv[0], tv[1], tv[2], tv[3]};
            *(LQ*)(lds + 32u * w + 16) = u32x4{tv[4], tv[5], tv[6], tv[7]};
            hib[w] = hb;
        }
        wave_fence();
    };
    auto kill_all = [&](uint32_t ref) {                                // every entry is more than 65535 bytes behind ref
        const uint32_t mk = (ref - 65536u) & 0x1FFFFu, lo2 = (mk & 0xFFFFu) * 0x10001u, hb = (mk >> 16) ? 0x55555555u : 0u;
        for (uint32_t i = (uint32_t)lane; i < 8192 / 16; i += kWave) ((__attribute__((address_space(3))) u32x4*)lds)[i] = u32x4{lo2, lo2, lo2, lo2};
        for (uint32_t i = (uint32_t)lane; i < 256; i += kWave) hib[i] = hb;
        wave_fence();
    };
    auto low_of = [&](uint32_t c) -> uint32_t { return LINKED ? ((ext && c >= b0) ? b0 : lowPos) : 0u; };   // where a catch-up from c stops
    if (!LINKED) {
        const uint32_t fill = (MODE == 2 && n >= 4) ? chk(ld32(src)) : 0u;
        const uint32_t bytes = MODE == 1 ? 8192u + 1024u : kTableBytes;
        for (uint32_t i = (uint32_t)lane; i < bytes / 16; i += kWave)
            ((__attribute__((address_space(3))) u32x4*)lds)[i] = u32x4{fill, fill, fill, fill};
        if (kChk1 && n >= 4 && chk(ld32(src))) {                      // the empty slots alias position 0: their check bit is position 0's
            wave_fence();
            for (uint32_t i = (uint32_t)lane; i < 256; i += kWave) hib[i] = 0xAAAAAAAAu;
        }
    }
    wave_fence();

    const bool limited = cap < lz4_bound(n - b0);         // lz4.c:2522
    const uint32_t mfl1 = n - kMfLimit + 1;               // mflimitPlusOne
    const uint32_t matchlimit = n - kLastLiterals;
    uint32_t ip = b0 + 1, anchor = b0;
    bool extra = false;                                   // next batch starts with the post-match probe
    // buffered sequence records [0, nrec): they describe src[f_anchor, anchor) and go to dst + f_op
    lbyte* recs = lds + (MODE == 1 ? 9216u : kTableBytes) + kSrcRing;
    uint32_t nrec = 0, f_op = 0, f_anchor = b0;
    // LINKED: where the output stands behind the sequences recorded so far.  A block that does not fit its room is stored raw by
    // the caller, but its table lives on: the reference gives up at the first sequence that does not fit (lz4.c:1139, :1215) with
    // the table as it was when that sequence's match was found, and the next block must find exactly that.  So the parser checks
    // every sequence's room itself, BEFORE the batch commits its table writes (the formatter's check is for the chunk kernels).
    uint32_t rop = 0;
    auto seq_size = [&](uint32_t lit, uint32_t mc) -> uint32_t { return 3u + (lit >= 15 ? (lit - 15) / 255 + 1 : 0u) + lit + (mc >= 15 ? (mc - 15) / 255 + 1 : 0u); };
    auto seq_fits = [&](uint32_t tp, uint32_t lit, uint32_t mc) -> bool {
        const uint32_t lb = lit >= 15 ? (lit - 15) / 255 + 1 : 0u;
        return !((uint64_t)tp + 1 + lit + (2 + 1 + kLastLiterals) + lit / 255 > cap || (uint64_t)tp + 1 + lb + lit + 2 + (1 + kLastLiterals) + (mc + 240) / 255 > cap);
    };
    bool overflow = false;
    typedef __attribute__((address_space(3))) uint64_t L64;
    typedef __attribute__((address_space(3))) u32x4 L128;
    typedef u32x4 __attribute__((aligned(1))) U128;
    typedef __attribute__((address_space(1))) U128 G128;
    auto rec_put = [&](uint32_t idx, uint32_t lit, uint32_t mcode, uint32_t off) {
        if (MODE == 1) *(L64*)(recs + 8u * idx) = (uint64_t)lit | ((uint64_t)mcode << 24) | ((uint64_t)off << 48);
        else *(L128*)(recs + 16u * idx) = u32x4{lit, mcode, off, 0u};
    };
    // Format the buffered sequences (lz4.c:1611-1740: token, literal length, literals, offset, match length), one per
    // lane; false = the output does not fit (limitedOutput, :1660-1668 and :1711-1716 - the parse never depends on the
    // output position, so checking a whole buffer at once returns 0 for exactly the inputs the reference returns 0 for).
    auto flush = [&]() -> bool {
#if defined(AOCLGPU_EXPERIMENTS) && defined(LZ4_EXP_NOFLUSH)   /* timing only: what the parser costs without the formatter (output is garbage) */
        f_anchor = anchor; nrec = 0; return true;
#endif
        const uint32_t L = (uint32_t)lane;
        const bool act = L < nrec;
        uint32_t lit = 0, mc = 0, off = 0;
        if (act) {
            if (MODE == 1) { const uint64_t r = *(L64*)(recs + 8u * L); lit = (uint32_t)r & 0xFFFFFFu; mc = (uint32_t)(r >> 24) & 0xFFFFFFu; off = (uint32_t)(r >> 48); }
            else { const u32x4 r = *(L128*)(recs + 16u * L); lit = r.x; mc = r.y; off = r.z; }
        }
        const uint32_t lb = lit >= 15 ? (lit - 15) / 255 + 1 : 0u, mb = mc >= 15 ? (mc - 15) / 255 + 1 : 0u;
        const uint32_t osz = act ? 3 + lb + lit + mb : 0u, isz = act ? lit + kMinMatch + mc : 0u;
        const uint32_t oinc = wave_scan_incl(osz), iinc = wave_scan_incl(isz);
        const uint32_t tp = f_op + oinc - osz;            // where this sequence's token goes
        const uint32_t sp = f_anchor + iinc - isz;        // where its literals are
        if (limited) {
            const bool bad = act && ((uint64_t)tp + 1 + lit + (2 + 1 + kLastLiterals) + lit / 255 > cap ||
                                     (uint64_t)tp + 1 + lb + lit + 2 + (1 + kLastLiterals) + (mc + 240) / 255 > cap);
            if (ballot(bad)) return false;
        }
        const uint32_t lp = tp + 1 + lb;                  // literals, then the offset at lp + lit
        // Literals of up to 64 bytes, one sequence per lane: four pieces per size class (16-byte pieces for 16 .. 64 bytes, 4-byte
        // pieces for 4 .. 15, single bytes below) at offsets 0, k, 2k clamped to lit - k and at lit - k: overlapping pieces carry the
        // same bytes, every access stays inside the run.  All loads of all classes are requested before the first store.
        const gbyte* s = src + sp; gbyte* d = dst + lp;
        const bool c16 = act && lit >= 16 && lit <= 64, c4 = act && lit >= 4 && lit < 16, c1 = act && lit >= 1 && lit < 4;
        const uint32_t e16 = lit - 16, p16a = e16 < 16 ? e16 : 16u, p16b = e16 < 32 ? e16 : 32u;
        const uint32_t e4 = lit - 4, p4a = e4 < 4 ? e4 : 4u, p4b = e4 < 8 ? e4 : 8u;
        u32x4 q0 = {0u, 0u, 0u, 0u}, q1 = q0, q2 = q0, q3 = q0;
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        uint8_t b0 = 0, b1 = 0, b2 = 0;
        if (c16) { q0 = *(const G128*)s; q1 = *(const G128*)(s + p16a); q2 = *(const G128*)(s + p16b); q3 = *(const G128*)(s + e16); }
        if (c4) { r0 = ld32(s); r1 = ld32(s + p4a); r2 = ld32(s + p4b); r3 = ld32(s + e4); }
        if (c1) { b0 = s[0]; b1 = s[lit >> 1]; b2 = s[lit - 1]; }
        if (act) {
            dst[tp] = (uint8_t)(((lit < 15 ? lit : 15) << 4) | (mc < 15 ? mc : 15));
            if (lit >= 15) dst[tp + lb] = (uint8_t)(lit - 15 - (lb - 1) * 255);
            st16(dst + lp + lit, off);
            if (mc >= 15) dst[lp + lit + 1 + mb] = (uint8_t)(mc - 15 - (mb - 1) * 255);
        }
        asm volatile("" ::: "memory");                    // (the loads above stay above)
        if (c16) { *(G128*)d = q0; *(G128*)(d + p16a) = q1; *(G128*)(d + p16b) = q2; *(G128*)(d + e16) = q3; }
        if (c4) { st32(d, r0); st32(d + p4a, r1); st32(d + p4b, r2); st32(d + e4, r3); }
        if (c1) { d[0] = b0; d[lit >> 1] = b1; d[lit - 1] = b2; }
        // the rare long fields, sequence after sequence by the whole wave: runs of 255 of a length, literals beyond 64 bytes
        for (uint64_t m = ballot(act && lb > 1); m; m &= m - 1) {
            const int k = (int)__builtin_ctzll(m);
            const uint32_t o = lane_get(tp, k) + 1, c = lane_get(lb, k) - 1;
            for (uint32_t i = L; i < c; i += kWave) dst[o + i] = 255;
        }
        for (uint64_t m = ballot(act && mb > 1); m; m &= m - 1) {
            const int k = (int)__builtin_ctzll(m);
            const uint32_t o = lane_get(lp + lit, k) + 2, c = lane_get(mb, k) - 1;
            for (uint32_t i = L; i < c; i += kWave) dst[o + i] = 255;
        }
        for (uint64_t m = ballot(act && lit > 64); m; m &= m - 1) {
            const int k = (int)__builtin_ctzll(m);
            wave_copy(dst + lane_get(lp, k), src + lane_get(sp, k), lane_get(lit, k), lane);
        }
        f_op += lane_get(oinc, 63); f_anchor = anchor; nrec = 0;
        wave_fence();
        return true;
    };
    // Source ring: the bytes around the parse position are staged in LDS as two 256-byte blocks (block b of the
    // chunk sits at ring offset (b & 1) * 256, the first 32 ring bytes are mirrored behind the end so that a 32-byte
    // read never wraps; block -1 is zeros: bytes before the chunk read as 0).  The block after the newest one is always
    // in flight in a register, requested ~256 bytes of parsing ahead of its use, so the 64 probe windows of a batch
    // are LDS reads instead of a global round trip.
    lbyte* ring = lds + (MODE == 1 ? 9216u : kTableBytes);
    int32_t rb = -4;                                      // newest block in the ring (blocks rb-1 and rb are valid)
    uint32_t pend = 0; bool have_pend = false;            // block rb+1, in flight
    auto blk_load = [&](int32_t b) -> uint32_t {
        uint32_t v = 0;
        if (b >= 0) {
            const uint32_t a = 256u * (uint32_t)b + 4u * (uint32_t)lane;
            if (a + 4 <= n) v = ld32(src + a);
            else if (a < n) for (uint32_t t = 0; t < n - a; ++t) v |= (uint32_t)src[a + t] << (8 * t);
        }
        return v;
    };
    auto blk_store = [&](int32_t b, uint32_t v) {
        *(L32*)(ring + (((uint32_t)b & 1u) << 8) + 4u * (uint32_t)lane) = v;
        if (!(b & 1) && lane < 8) *(L32*)(ring + 512u + 4u * (uint32_t)lane) = v;
    };

    if (n - b0 >= kMinLength) {
        if (LINKED) {                                     // the block's first byte (:1042; a chunk's is the zero-filled table's position 0)
            const uint64_t s8 = ld64(src + b0);
            if (lane == 0) tput_blind(lz4_hash2<U16>((uint32_t)s8, (uint32_t)(s8 >> 32)), b0 + idxBase);
            wave_fence();
        }
        uint32_t i0 = 0;                                  // probes of the current search already examined
        for (;;) {
            asm volatile("; batch begin" ::: "memory");
            const uint32_t ex = extra ? 1u : 0u;
            const uint32_t s0 = uni(ip + i0);             // position of lane 0 while the schedule's steps are 1 byte
            if (kChk1 && n > 0x20000u && (i0 > 66u || ip + 80u - sweep_ref >= kSweepEvery)) {   // (i0 <= 66: the lowest read is at most ip + 66)
                // the batch reads the table at the positions of probes i0 - 1 .. i0 + 63 of the search that began at ip; every later
                // batch reads at or behind this one's lowest position
                const uint32_t lo = ip + sched(i0 > 1 ? i0 - 2 : 0, accel) - 2;
                if (lo - sweep_ref >= kSweepEvery) { age_sweep(lo, 0); sweep_ref = lo; }
            }
            bool done = false;
            int special = -1;                             // lane whose sequence takes the one-at-a-time path below
            uint32_t pos, cand, fw, bk4; bool mf;         // per lane, read by that path
            if (accel == 1 && i0 <= 65 + ex && s0 + 92 <= mfl1) {
                // ---------------- 64 CONSECUTIVE positions s0 + lane (lz4.c:1483-1608, 1746-1804) ----------------
                // Probe j of a search sits at its start + j for j <= 65 (step = searchMatchNb++ >> 6), so the first `own`
                // lanes are probes of the current search and EVERY lane is a probe of any search that starts behind a match
                // of this batch: the batch resolves sequence after sequence.  Far from the chunk's end (s0 + 92 <=
                // mflimitPlusOne) every lane is a valid probe, every window is inside the chunk and no match reaches
                // matchlimit; the chunk's last hundred bytes take the general path.
                const uint32_t own = 66 + ex - i0 < 64 ? 66 + ex - i0 : 64;
                pos = s0 + (uint32_t)lane;
                const int32_t w0 = (int32_t)s0 - 4;       // the ring must hold [w0, w0 + 96)
                if ((uint32_t)(w0 - 256 * (rb - 1)) > 512u - 96u) {
                asm volatile("; ring refill begin");
                    const int32_t bl = w0 >> 8, bh = (w0 + 95) >> 8;
                    if (have_pend && rb + 1 == bh && bl >= rb) { blk_store(rb + 1, pend); rb = (int32_t)uni((uint32_t)(rb + 1)); }
                    else {
                        const uint32_t v0 = blk_load(bl), v1 = blk_load(bh);
                        blk_store(bl, v0);
                        if (bh != bl) blk_store(bh, v1);
                        rb = (int32_t)uni((uint32_t)bh);
                    }
                    pend = blk_load(rb + 1); have_pend = true;
                    wave_fence();
                }
                asm volatile("; ring refill end");
                // 32-byte probe windows [p-4, p+28): a = {p-4, p, p+4, p+8}, b = {p+12, p+16, p+20, p+24}
                const uint32_t roff = (uint32_t)(w0 + lane) & 511u;
                const u32x4 a = lds_ld128(ring + roff), b = lds_ld128(ring + roff + 16);
                if (extra && i0 == 0) {                   // back-fill ip-2 (lz4.c:1750) before the batch reads the table: lane 0's window holds its bytes
                    const uint32_t f_lo = __builtin_amdgcn_alignbit(a.y, a.x, 16), f_hi = a.y >> 16;
                    const uint32_t h2 = lz4_hash2<U16>(f_lo, f_hi);
                    if (lane == 0) tput_blind(h2, entry_of(ip - 2, chk(f_lo)));
                    wave_fence();
                }
                STAGE(0);
                const uint32_t my4 = a.y;
                const uint32_t h = lz4_hash2<U16>(a.y, a.z);
                const uint32_t ent = tget(h);
                cand = cand_of(ent, pos);
                const uint32_t mychk = chk(my4);
                // only lanes that can still pass the test fetch their candidate window
                const bool plausible = usable(ent, cand, pos, mychk);
                STAGE(1);
                // candidate windows [c-4, c+28) (cand < pos, so they end inside the chunk): 4 bytes of catch-up, the 4 bytes of
                // the match test and 24 bytes of forward extension arrive in one round trip
                u32x4 A = {~a.x, ~a.y, 0u, 0u}, B = {0u, 0u, 0u, 0u};
                if (plausible) {
                    const uint32_t cq = cand >= 4 ? cand - 4 : 0u;
                    A = *(const G128*)(src + cq); B = *(const G128*)(src + cq + 16);
                }
                // exact first in-batch slot reuse: per-slot minimum lane by write / read-back rounds
                wave_fence();
                const uint32_t w = lds_first_lane_per_slot((uint32_t)(uintptr_t)&table[h], (uint32_t)lane);
                wave_fence();
                const uint64_t dups = ballot(w != (uint32_t)lane);
                table[h] = (T)ent;                        // undo the lane ids (the low halves; extra bits were not touched)
                wave_fence();
                if (ballot(plausible && cand < 4)) {      // window that starts before the chunk: loaded from byte 0, moved up 4 - cand bytes
                    if (plausible && cand < 4) {
                        const uint32_t k = 8u * cand;     // alignbit(W[i], W[i-1], 32 - 8 * (4 - cand))
                        B.w = __builtin_amdgcn_alignbit(B.w, B.z, k); B.z = __builtin_amdgcn_alignbit(B.z, B.y, k); B.y = __builtin_amdgcn_alignbit(B.y, B.x, k); B.x = __builtin_amdgcn_alignbit(B.x, A.w, k);
                        A.w = __builtin_amdgcn_alignbit(A.w, A.z, k); A.z = __builtin_amdgcn_alignbit(A.z, A.y, k); A.y = __builtin_amdgcn_alignbit(A.y, A.x, k);
                        A.x = __builtin_amdgcn_alignbit(A.x, 0u, k);
                    }
                }
                STAGE(2);
                // Lanes whose slot an earlier lane of the batch also names ("duplicates") have a candidate that depends on the
                // parse: the earlier lane's position if that lane's table write happens, else the table entry.  Such a lane is
                // still decided when NEITHER can match - its table candidate fails the 4-byte test and no earlier lane on its slot
                // holds the same 4 bytes: it is a probe that finds nothing, whoever its candidate is.  The batch is cut at the
                // first duplicate that could match.  `later` = this lane's slot is named again by these later lanes (the last
                // one whose write happens owns the slot when the batch commits).
                // One kind of duplicate that could match IS decided (round 4, second half): the lane whose LATEST earlier lane on the
                // slot holds the same 4 bytes - a repeat at a distance below 64, the usual reason two lanes of a batch share a slot
                // (binary and record data are full of them).  Its candidate is that lane's position, provided that lane's table write
                // happens, i.e. that lane is a probe and not a position inside one of the batch's matches - which the walk knows when
                // it arrives there.  Such a "conditional" lane gets its candidate window from the source ring (it is the other
                // lane's probe window), stops the walk like a special lane, and is either taken (the walk goes on from it) or, if
                // its candidate lane turns out to lie inside a match, cuts the batch there as every could-match duplicate used to.
                const uint64_t tabok = ballot(plausible && A.y == my4);
                uint32_t lim = 64;
                uint64_t condm = dups;
                if (LINKED && dups) { lim = (uint32_t)__builtin_ctzll(dups); condm = 0; }       // (linked blocks keep the plain rule: cut at a duplicate)
#if defined(AOCLGPU_EXPERIMENTS) && defined(LZ4_EXP_NOCUT)   /* timing only: what in-batch duplicates cost - every duplicate is an ordinary lane (wrong parse) */
                condm = 0;
#endif
                LZ4_CC(0, (uint32_t)__builtin_popcountll(dups));
                LZ4_CC(5, lim < 64 ? 1u : 0u); LZ4_CC(6, lim); LZ4_CC(7, 1);
                const bool isc = condm != 0 && __builtin_amdgcn_inverse_ballot_w64(condm);   // (stops the walk: resolved when it arrives)
                const uint64_t okmask = (tabok | condm) & (lim >= 64 ? ~0ull : ((1ull << lim) - 1));   // lanes < lim are decided exactly as the sequential parse decides them
                // forward extension (<= 24 bytes, LZ4_count lz4.c:656-679) and catch-up (<= 4 bytes), in registers
                auto counts = [&](const u32x4& A_, const u32x4& B_, uint32_t& fw_, uint32_t& bk_) {
                    const uint64_t x1 = (uint64_t)(a.z ^ A_.z) | ((uint64_t)(a.w ^ A_.w) << 32);
                    const uint64_t x2 = (uint64_t)(b.x ^ B_.x) | ((uint64_t)(b.y ^ B_.y) << 32);
                    const uint64_t x3 = (uint64_t)(b.z ^ B_.z) | ((uint64_t)(b.w ^ B_.w) << 32);
                    fw_ = x1 ? ((uint32_t)__builtin_ctzll(x1) >> 3) : 8;
                    if (fw_ == 8) fw_ += x2 ? ((uint32_t)__builtin_ctzll(x2) >> 3) : 8;
                    if (fw_ == 16) fw_ += x3 ? ((uint32_t)__builtin_ctzll(x3) >> 3) : 8;
                    const uint32_t yb = a.x ^ A_.x;
                    bk_ = yb ? ((uint32_t)__builtin_clz(yb) >> 3) : 4;
                };
                counts(A, B, fw, bk4);
                mf = fw == 24;
                if (PROF) { cnt[0]++; cnt[6] += lim; if (i0) cnt[3]++; }
                STAGE(3);
                asm volatile("; resolve begin" ::: "memory");   // scheduling fence: keeps the batch's loads/LDS traffic above the walk
                // ---------------- resolve every sequence this batch holds ----------------
                // Every matching lane precomputes where the chain continues after its match (`nxt` = first matching lane at or
                // after the match's end, >= 64: none), so the scalar walk is one readlane per sequence; everything else (the
                // anchor of every sequence, the table commit, the records) is derived from 64-bit lane masks once per batch.
                // Sequences that need a loop (longer extension / catch-up than the windows hold) are "special": they end
                // the walk (their `nxt` is negative) and take the one-at-a-time path below.
                uint32_t endl = (uint32_t)lane + kMinMatch + fw;                             // match end, in lanes (<= 91)
                uint32_t nxt;
                {
                    const uint64_t t = (okmask >> (endl < 63 ? endl : 63)) | (1ull << 63);   // (endl >= 4: the shift clears bit 63)
                    nxt = endl + (uint32_t)__builtin_ctzll(t);
                }
                uint64_t marks = 0;                           // lanes that start a sequence
                {
                    const uint64_t first = okmask & (own >= 64 ? ~0ull : ((1ull << own) - 1));
                    if (first) {                              // marks |= 1 << mw; mw = nxt[mw] until mw >= 64 or a lane that stops the walk (lz_group.hip.h)
                        uint32_t mw, stop, start = (uint32_t)__builtin_ctzll(first);
                        uint32_t nxw = (mf || isc) ? 0x80000000u : nxt;
                        for (;;) {
                            uint64_t mk;
                            lz_walk_tokens(nxw, mk, mw, stop, start);
                            marks |= mk;
                            if (!stop) break;
                            if (!((condm >> mw) & 1)) { special = (int)mw; break; }
                            // A duplicate lane: its candidate is the position of the LATEST earlier lane on its slot whose table write has
                            // happened - as a probe, or as the ip-2 back-fill of the match that covers it - else the table's entry.
                            uint32_t il = 64;
                            {
                                uint64_t L = ballot(h == lane_get(h, (int)mw)) & ((1ull << mw) - 1);
                                while (L) {
                                    const uint32_t x = 63u - (uint32_t)__builtin_clzll(L);
                                    const uint64_t lowm = marks & ((2ull << x) - 1);             // matches that start at or below it (x <= 62)
                                    bool wr = true;
                                    if (lowm) { const int mm = 63 - (int)__builtin_clzll(lowm); const uint32_t em = lane_get(endl, mm); wr = (uint32_t)mm == x || em <= x || em == x + 2; }
                                    if (wr) { il = x; break; }
                                    L &= ~(1ull << x);
                                }
                            }
                            const bool written = il < 64;
                            const bool hit = written ? lane_get(my4, (int)il) == lane_get(my4, (int)mw) : ((tabok >> mw) & 1) != 0;
                            LZ4_CC(written ? 2 : 3, 1); if (!hit) LZ4_CC(4, 1);
                            uint32_t from;                                                        // where the walk goes on
                            if (hit) {
                                if (written) {                        // that lane's probe window (from the source ring) is the candidate window
                                    if ((uint32_t)lane == mw) {
                                        const uint32_t rc = (uint32_t)(w0 + (int32_t)il) & 511u;
                                        const u32x4 Ac = lds_ld128(ring + rc), Bc = lds_ld128(ring + rc + 16);
                                        counts(Ac, Bc, fw, bk4);
                                        mf = fw == 24; cand = s0 + il; endl = (uint32_t)lane + kMinMatch + fw;
                                    }
                                }
                                if (lane_get((uint32_t)mf, (int)mw)) { special = (int)mw; break; }   // taken, but its match leaves the window
                                const uint32_t e = lane_get(endl, (int)mw);
                                const uint64_t t = (okmask >> (e < 63 ? e : 63)) | (1ull << 63);
                                const uint32_t nx1 = e + (uint32_t)__builtin_ctzll(t);
                                nxw = (uint32_t)lane == mw ? nx1 : nxw;                          // taken: the walk goes on from it
                                from = mw;
                            } else {
                                // a probe that finds nothing: the search goes on at the next matching lane (the first search of the batch
                                // only owns `own` lanes)
                                const uint64_t rest = okmask & (marks == 0 && own < 64 ? ((1ull << own) - 1) : ~0ull) & ~((2ull << mw) - 1);
                                if (!rest) break;
                                from = (uint32_t)__builtin_ctzll(rest);
                            }
                            start = from;
                        }
                    }
                }
                STAGE2(0);
                // per lane: the anchor of the sequence it belongs to = end of the last match that starts below it
                const int32_t arel0 = (int32_t)(anchor - s0);       // <= 0; negative: literals carried in from earlier batches
                bool marked = __builtin_amdgcn_inverse_ballot_w64(marks);
                const uint32_t qend = wave_shr1(wave_scan_max(marked ? endl : 0u));          // 0: no match starts below this lane
                const int32_t la = qend ? (int32_t)qend : arel0;
                const uint32_t room = (uint32_t)((int32_t)lane - la);                        // m_pos - anchor
                const uint32_t cl = cand - low_of(cand);                                      // room for a catch-up on the candidate's side
                const uint32_t maxback = room < cl ? room : cl;
                {   // a catch-up that runs past the 4 bytes of the window needs the loop: cut the chain in front of it
                    const uint64_t need = ballot(marked && bk4 == 4 && maxback > 4);
                    if (need) { special = (int)__builtin_ctzll(need); marks &= ~(~0ull << special); marked = __builtin_amdgcn_inverse_ballot_w64(marks); }
                }
                if (LINKED && marks) {                        // the first sequence that does not fit ends the block, with the table as of its match
                    const uint32_t bk = bk4 < maxback ? bk4 : maxback, lit = room - bk, mc = bk + fw;
                    const uint32_t osz = marked ? seq_size(lit, mc) : 0u, oinc = wave_scan_incl(osz);
                    const uint64_t bad = ballot(marked && !seq_fits(rop + oinc - osz, lit, mc));
                    if (bad) { overflow = true; special = (int)__builtin_ctzll(bad); marks &= ~(~0ull << special); marked = __builtin_amdgcn_inverse_ballot_w64(marks); }
                    else rop += lane_get(oinc, 63);
                }
                const uint64_t cover = ballot(marked || (uint32_t)lane < qend);               // lanes inside a match [m, endl) (below `special`: of the sequences kept)
                uint32_t cur = 0;                             // first lane of the unfinished search
                if (marks) cur = lane_get(endl, 63 - (int)__builtin_clzll(marks));
                STAGE2(1);
                {   // table writes the reference performs: every probed lane of every search, plus the ip-2 back-fills
                    const uint32_t R = special >= 0 ? (uint32_t)special + 1 : (marks == 0 ? (own < lim ? own : lim) : (cur >= lim ? cur : lim));
                    const uint64_t belowR = R >= 64 ? ~0ull : ~(~0ull << R);
                    const uint64_t ends = (cover << 1) & (~cover | marks);                    // lanes where a match ends
                    const uint64_t commit = (belowR & ~(cover & ~marks)) | ((ends & belowR) >> 2);
                    // several writers of one slot: the latest of them owns it (rounds of "the largest id sticks" over the slots'
                    // low halves, ids = lane + 1 over a zero)
                    if (dups) {
                        typedef __attribute__((address_space(3))) uint16_t LU16;
                        const bool cm = __builtin_amdgcn_inverse_ballot_w64(commit);
                        if (cm) *(LU16*)&table[h] = 0;
                        wave_fence();
                        const uint32_t top = lds_last_lane_per_slot((uint32_t)(uintptr_t)&table[h], (uint32_t)lane + 1u, cm);
                        wave_fence();
                        if (cm && top == (uint32_t)lane + 1u) tput(h, entry_of(pos, mychk), ent);
                    }
                    else if (__builtin_amdgcn_inverse_ballot_w64(commit)) tput(h, entry_of(pos, mychk), ent);
                }
                wave_fence();
                STAGE(4);
                if (marks) {
                    // per-sequence quantities, in the lane that found the match (lz4.c:1611-1740); one record per sequence, in
                    // sequence order behind the buffered ones
                    const uint32_t back = bk4 < maxback ? bk4 : maxback;
                    const uint32_t idx = nrec + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(marks >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)marks, 0u));
                    if (marked) rec_put(idx, room - back, back + fw, pos - cand);
                    const uint32_t ns = (uint32_t)__builtin_popcountll(marks);
                    nrec += ns;
                    if (PROF) cnt[1] += ns;
                    ip = s0 + cur; anchor = ip; extra = true; i0 = 0;
                }
                STAGE(5);
                if (special < 0) {
                    if (marks == 0) i0 += own < lim ? own : lim;       // the search goes on
                    else if (cur < lim) i0 = lim - cur;                // lanes [cur, lim) of the next search found nothing
                }
            } else {
                asm volatile("; general batch begin");
                // ---------------- general batch: probe i0 + lane of the schedule (any acceleration, long searches, the chunk's
                // end); lane k evaluates probe k speculatively, the first hit below the first in-batch slot reuse is the match
                // (lz4.c:1483-1608), resolved one at a time ----------------
                const uint32_t base = extra ? ip + 1 : ip;
                const int32_t j = (int32_t)(i0 + (uint32_t)lane) - (int32_t)ex;             // j = -1: the post-match probe at ip
                const uint64_t p64 = j < 0 ? (uint64_t)base - 1 : (uint64_t)base + sched((uint32_t)j, accel);
                pos = p64 > 0x7FFFFF00ull ? 0x7FFFFF00u : (uint32_t)p64;
                const bool valid = (uint64_t)base + sched((uint32_t)(j + 1), accel) <= mfl1;   // lz4.c:1513
                Win pw = {0, 0};
                if (valid) pw = load_win(src, pos);
                if (extra && i0 == 0) {                   // back-fill ip-2 (lz4.c:1750) before the batch reads the table
                    const uint64_t fb = ld64(src + ip - 2);
                    const uint32_t h2 = lz4_hash2<U16>((uint32_t)fb, (uint32_t)(fb >> 32));
                    if (lane == 0) tput_blind(h2, entry_of(ip - 2, chk((uint32_t)fb)));
                    wave_fence();
                }
                const uint32_t my4 = (uint32_t)(pw.lo >> 32);
                const uint32_t h = lz4_hash2<U16>(my4, (uint32_t)pw.hi);
                uint32_t ent = 0;
                if (valid) ent = tget(h);
                cand = cand_of(ent, pos);
                const uint32_t mychk = chk(my4);
                const bool plausible = valid && usable(ent, cand, pos, mychk);
                Win cw = {~pw.lo, 0};
                if (plausible) cw = load_win(src, cand);
                wave_fence();
                const uint32_t w = lds_first_lane_per_slot((uint32_t)(uintptr_t)&table[h], (uint32_t)lane, valid);
                wave_fence();
                if (valid) table[h] = (T)ent;
                wave_fence();
                const uint32_t e = (uint32_t)first_lane(ballot(!valid));
                const uint64_t tabok = ballot(plausible && (uint32_t)(cw.lo >> 32) == my4);
                uint64_t later = 0;
                uint32_t f = 64;
                for (uint64_t d = ballot(valid && w != (uint32_t)lane); d; d &= d - 1) {
                    const int k = (int)__builtin_ctzll(d);
                    const uint64_t before = (1ull << k) - 1;
                    const uint64_t same = ballot(valid && h == lane_get(h, k)) & before;
                    if (((tabok >> k) & 1) || (ballot(my4 == lane_get(my4, k)) & same) != 0) { f = (uint32_t)k; break; }
                    if ((same >> lane) & 1) later |= 1ull << k;
                }
                const uint32_t lim = f < e ? f : e;
                const uint64_t okmask = tabok & (lim >= 64 ? ~0ull : ((1ull << lim) - 1));
                {
                    const uint64_t x1 = pw.hi ^ cw.hi;
                    fw = x1 ? ((uint32_t)__builtin_ctzll(x1) >> 3) : 8;
                    const uint32_t maxfwd = matchlimit - (pos + kMinMatch);
                    mf = fw == 8 && maxfwd > 8;
                    fw = fw < maxfwd ? fw : maxfwd;
                    const uint32_t yb = (uint32_t)pw.lo ^ (uint32_t)cw.lo;
                    bk4 = yb ? ((uint32_t)__builtin_clz(yb) >> 3) : 4;
                }
                if (PROF) { cnt[0]++; cnt[2]++; cnt[6] += lim; }
                if (okmask) special = (int)__builtin_ctzll(okmask);
                {
                    const uint32_t R = special >= 0 ? (uint32_t)special + 1 : lim;
                    const uint64_t commit = R >= 64 ? ~0ull : ~(~0ull << R);
                    if (__builtin_amdgcn_inverse_ballot_w64(commit) && (later & commit) == 0) tput(h, entry_of(pos, mychk), ent);
                }
                wave_fence();
                if (special < 0) {
                    if (e < 64 && e <= f) done = true;                                        // next probe would pass mflimitPlusOne
                    else i0 += lim;
                }
            }
            asm volatile("; general batch end");
            if (LINKED && overflow) return 0;
            if (special >= 0) {
                asm volatile("; special begin");
                STAGE(7);                                 // (a general batch)
                // ---------------- one special sequence, lane `special` (lz4.c:1611-1740) ----------------
                const int m = special;
                const uint32_t m_pos = lane_get(pos, m), m_cand = lane_get(cand, m);
                uint32_t back, fwd = lane_get(fw, m);
                const bool more_fwd = lane_get((uint32_t)mf, m) != 0;
                {   // the first round of the longer catch-up (64 bytes back) and of the longer extension (256 bytes forward)
                    // is issued together: one global round trip instead of two
                    const uint32_t a = m_pos - anchor, mcl = m_cand - low_of(m_cand), maxback = a < mcl ? a : mcl;
                    const uint32_t b4 = lane_get(bk4, m);
                    back = b4 < maxback ? b4 : maxback;
                    const bool more_back = b4 == 4 && maxback > 4;
                    bool eqb = false;
                    uint32_t x = 0xFFFFFFFFu;
                    const uint32_t fa = m_pos + kMinMatch, fb = m_cand + kMinMatch;
                    if (more_back && back + (uint32_t)lane < maxback) eqb = src[m_pos - 1 - back - (uint32_t)lane] == src[m_cand - 1 - back - (uint32_t)lane];
                    if (more_fwd) {
                        const uint32_t o = fwd + 4u * (uint32_t)lane, pa = fa + o;
                        if (pa < matchlimit) {
                            x = ld32(src + pa) ^ ld32(src + fb + o);
                            const uint32_t avail = matchlimit - pa;
                            if (avail < 4) x |= 0xFFFFFFFFu << (8 * avail);
                        }
                    }
                    if (more_back) {
                        int c = first_lane(ballot(!eqb));
                        back += (uint32_t)c;
                        while (c == 64 && back < maxback) {
                            const uint32_t k = back + (uint32_t)lane;
                            bool eq = false;
                            if (k < maxback) eq = src[m_pos - 1 - k] == src[m_cand - 1 - k];
                            c = first_lane(ballot(!eq));
                            back += (uint32_t)c;
                        }
                    }
                    if (more_fwd) {
                        for (;;) {
                            const int l = first_lane(ballot(x != 0));
                            if (l < 64) { fwd += 4u * (uint32_t)l + ((uint32_t)__builtin_ctz(lane_get(x, l)) >> 3); break; }
                            fwd += 256;
                            const uint32_t o = fwd + 4u * (uint32_t)lane, pa = fa + o;
                            x = 0xFFFFFFFFu;
                            if (pa < matchlimit) {
                                x = ld32(src + pa) ^ ld32(src + fb + o);
                                const uint32_t avail = matchlimit - pa;
                                if (avail < 4) x |= 0xFFFFFFFFu << (8 * avail);
                            }
                        }
                    }
                }
                const uint32_t mstart = m_pos - back;
                const uint32_t mcode = back + fwd;
                if (LINKED) {
                    if (!seq_fits(rop, mstart - anchor, mcode)) return 0;
                    rop += seq_size(mstart - anchor, mcode);
                }
                if (lane == 0) rec_put(nrec, mstart - anchor, mcode, m_pos - m_cand);
                nrec++;
                if (PROF) { cnt[1]++; cnt[4]++; }
                ip = mstart + kMinMatch + mcode;
                anchor = ip;
                if (kChk1 && n > 0x20000u && ip - sweep_ref > kJumpMax) {          // a long step: see age_sweep
                    if (ip - m_pos > kDistMax + kSweepAhead) kill_all(ip);          // (every entry is at most a few bytes behind m_pos)
                    else age_sweep(m_pos, ip - m_pos);                              // what is dead at ip goes now, while ages are unambiguous
                    sweep_ref = ip;
                }
                if (ip >= mfl1) done = true;              // lz4.c:1747
                else { extra = true; i0 = 0; }
                STAGE(6);
                asm volatile("; special end");
            }
            asm volatile("; resolve end" ::: "memory");
            if (nrec > kRecFlush || done) { asm volatile("; flush begin"); wave_fence(); if (PROF) cnt[5]++; if (!flush()) return 0; asm volatile("; flush end"); }
            STAGE(7);
            if (done) break;
        }
    }
    // ---------------- last literals (lz4.c:1808-1834) ----------------
    uint32_t op = f_op;
    {
        const uint32_t lastRun = n - anchor;
        *last_run = lastRun;
        if (limited && (uint64_t)op + lastRun + 1 + (lastRun + 255 - 15) / 255 > cap) return 0;
        if (lane == 0) dst[op] = (uint8_t)((lastRun < 15 ? lastRun : 15) << 4);
        op++;
        if (lastRun >= 15) op += put_len_bytes(dst + op, lastRun - 15, lane);
        wave_copy(dst + op, src + anchor, lastRun, lane);
        op += lastRun;
    }
#ifdef AOCLGPU_EXPERIMENTS
    if (PROF && lane == 0) { for (int k = 0; k < 8; ++k) atomicAdd(&g_lz4_prof[k], (unsigned long long)tacc[k]); for (int k = 0; k < 8; ++k) atomicAdd(&g_lz4_prof[8 + k], (unsigned long long)cnt[k]); }
    if (lane == 0) for (int k = 0; k < 8; ++k) if (ccnt[k]) atomicAdd(&g_lz4_cond[k], (unsigned long long)ccnt[k]);
#ifdef LZ4_PROF_SPLIT
    if (PROF && lane == 0) for (int k = 0; k < 8; ++k) atomicAdd(&g_lz4_prof2[k], (unsigned long long)tacc2[k]);
#endif
#endif
#undef STAGE2
#undef STAGE
    return (int)op;
}

// ------------------------------------------------------------------------------------------------
// Safe decoder: accepts exactly the streams the reference's safe loop accepts (lz4.c:3563-3790).
//
// Two paths over the same state (ip, op):
//  * window path: the compressed bytes are staged through a 2 KB LDS ring; lane k parses "a token at ip+k"
//    (token, literal length, offset), a scalar walk over the lanes picks the real token starts, prefix sums give
//    every sequence its output position, then all literals and all matches whose source lies before the window
//    are copied at once (one global round trip for the whole window); matches that read this window's own
//    output follow in order.  Only sequences without length-extension bytes and far from both buffer ends take
//    this path, so every bound the reference checks is implied;
//  * sequential path (one sequence per call): everything else, including every error and the last sequence.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kRing = 2048, kRingMask = kRing - 1;

// one sequence, all rules of lz4.c:3563-3790.  returns 0 = continue, 1 = block finished, <0 = malformed
__device__ int lz4_decode_one(const gbyte* __restrict__ src, const int32_t iend, gbyte* dst, const int32_t oend,
                              int32_t& ip, int32_t& op, int lane)
{
    const uint32_t token = uni(src[ip]);
    ip++;
    int64_t len = token >> 4;
    if (len == 15) {                                      // read_variable_length (lz4.c:3331-3352)
        if (ip >= iend - 15) return -ip - 1;
        uint32_t s;
        do {
            s = uni(src[ip]); ip++; len += s;
            if (ip >= iend - 15) break;
        } while (s == 255);
    }
    if (op + len > (int64_t)oend - (int64_t)kMfLimit || ip + len > (int64_t)iend - (2 + 1 + (int64_t)kLastLiterals)) {
        if (ip + len != iend || op + len > oend) return -ip - 1;        // lz4.c:3654-3664
        wave_copy(dst + op, src + ip, (uint32_t)len, lane);
        op += (int32_t)len;
        return 1;
    }
    wave_copy(dst + op, src + ip, (uint32_t)len, lane);
    ip += (int32_t)len; op += (int32_t)len;
    const uint32_t offset = uni(ld16(src + ip));
    ip += 2;
    int64_t ml = token & 15;
    if (ml == 15) {
        uint32_t s;
        do {
            s = uni(src[ip]); ip++; ml += s;
            if (ip >= iend - (int32_t)kLastLiterals + 1) return -ip - 1;
        } while (s == 255);
    }
    ml += kMinMatch;
    if ((int64_t)offset > op) return -ip - 1;                               // offset outside the block
    if (op + ml > (int64_t)oend - (int64_t)kLastLiterals) return -ip - 1;   // last 5 bytes are literals
    // match copy; every source byte lies before `op`, so (i mod offset) never reads a byte this
    // sequence has not produced yet within one 64-lane round
    wave_fence();
    gbyte* d = dst + op;
    const uint32_t mlen = (uint32_t)ml;
    if (offset == 0) {                                    // lz4.c:3745-3753: run of zeros
        for (uint32_t i = (uint32_t)lane; i < mlen; i += kWave) d[i] = 0;
    } else if (offset >= kWave) {
        const gbyte* s = d - offset;
        for (uint32_t r = 0; r < mlen; r += kWave) {
            const uint32_t i = r + (uint32_t)lane;
            uint8_t v = 0;
            if (i < mlen) v = s[i];
            if (i < mlen) d[i] = v;
            wave_fence();
        }
    } else {
        const gbyte* s = d - offset;
        uint32_t k = (uint32_t)lane % offset;
        const uint32_t adv = kWave % offset;
        for (uint32_t i = (uint32_t)lane; i < mlen; i += kWave) {
            d[i] = s[k];
            k += adv; if (k >= offset) k -= offset;
        }
    }
    wave_fence();
    op += (int32_t)mlen;
    return 0;
}

// One sequence of a RAP partition that is not the last one (AOCL_LZ4_decompress_generic_mt with is_last_thread == 0,
// lz4.c:3807-4304): the partition holds whole sequences only, so it may end right after a match, a match may run up to
// the very end of the partition's output, and the end-of-block literal rules do not apply.  Every access is bounds
// checked against the partition.  returns 0 = continue, 1 = partition finished, <0 = malformed
__device__ int lz4_decode_one_rap(const gbyte* __restrict__ src, const int32_t iend, gbyte* dst, const int32_t oend,
                                  int32_t& ip, int32_t& op, int lane)
{
    const uint32_t token = uni(src[ip]);
    ip++;
    int64_t len = token >> 4;
    if (len == 15) {
        uint32_t s;
        do {
            if (ip >= iend) return -ip - 1;
            s = uni(src[ip]); ip++; len += s;
        } while (s == 255);
    }
    if (ip + len > (int64_t)iend || op + len > (int64_t)oend) return -ip - 1;
    wave_copy(dst + op, src + ip, (uint32_t)len, lane);
    ip += (int32_t)len; op += (int32_t)len;
    if (ip == iend) return 1;                             // ended with literals
    if (ip + 2 > iend) return -ip - 1;
    const uint32_t offset = uni(ld16(src + ip));
    ip += 2;
    int64_t ml = token & 15;
    if (ml == 15) {
        uint32_t s;
        do {
            if (ip >= iend) return -ip - 1;
            s = uni(src[ip]); ip++; ml += s;
        } while (s == 255);
    }
    ml += kMinMatch;
    if (offset == 0 || (int64_t)offset > op || op + ml > (int64_t)oend) return -ip - 1;
    wave_fence();
    gbyte* d = dst + op;
    const gbyte* sp = d - offset;
    const uint32_t mlen = (uint32_t)ml;
    if (offset >= kWave) {
        for (uint32_t r = 0; r < mlen; r += kWave) {
            const uint32_t i = r + (uint32_t)lane;
            uint8_t v = 0;
            if (i < mlen) v = sp[i];
            if (i < mlen) d[i] = v;
            wave_fence();
        }
    } else {
        uint32_t k = (uint32_t)lane % offset;
        const uint32_t adv = kWave % offset;
        for (uint32_t i = (uint32_t)lane; i < mlen; i += kWave) { d[i] = sp[k]; k += adv; if (k >= offset) k -= offset; }
    }
    wave_fence();
    op += (int32_t)mlen;
    return ip == iend ? 1 : 0;
}

// bytes [o, o+8) of the 24-byte little-endian string v0|v1|v2 (o <= 16)
__device__ __forceinline__ uint64_t ext64(uint64_t v0, uint64_t v1, uint64_t v2, uint32_t o)
{
    const uint64_t a = o < 8 ? v0 : v1, b = o < 8 ? v1 : v2;
    const uint32_t sh = (o & 7) * 8;
    return sh ? (a >> sh) | (b << (64 - sh)) : a;
}

// op0 > 0: dst[0, op0) is history the block may copy from (linked blocks of an LZ4 frame, LZ4_decompress_safe_usingDict
// with the dictionary in front of the output, lz4.c:4985-5010); the block's own output is dst[op0, op0 + cap) and the
// return value is the end position op0 + decoded size.
// (HIST is a template parameter and the function is inlined into each kernel: a shared, non-inlined copy receives its
//  arguments in VGPRs, loses its scalar control flow, and the plain decoder drops from 250 to 156 GiB/s.)
//
// Group path (round 4).  The tokens of a block are a chain (where the next token starts is only known once this one is
// parsed), everything else is not.  So the wave SCANS - lane k parses "a token starts at ip + k" from the staged input, a
// scalar walk over the lanes picks the real starts - and only RECORDS what it finds (literal source, literal length, match
// length, offset: 8 bytes per sequence in LDS).  When 64 sequences are buffered, or their output reaches 2 KB, or a token
// comes up that the scan does not take (a length of 255 or more in one field, more than 64 literal or match bytes, the
// buffer ends), the group is EXECUTED with one sequence per lane: a prefix sum gives every sequence its place in a 2 KB
// output block kept in LDS; all literals and all matches whose source lies before the block are fetched in one round of
// loads (four pieces per size class, every access inside its run) and written into the block; matches that read the block
// itself - or the 64 bytes in front of it, kept as history in front of the block - follow in sequence order as LDS-to-LDS
// copies (130 cycles each instead of a trip to memory); the block then leaves with 16-byte stores.  Whatever the scan does
// not take goes through the sequential one-sequence path, which implements every check of the reference.
#ifndef LZ4D_RECS
#define LZ4D_RECS 96        // round 5: more records than lanes - a group runs with all 64 lanes busy (45 of 64 with a buffer of 64)
#endif
constexpr uint32_t kDecRecs = LZ4D_RECS, kDecBlock = kGrpBlock, kDecHist = kGrpHist;
constexpr uint32_t kDecLds = kRing + 16 + kDecRecs * 4 + kGrpLds;

template <bool RAP, bool HIST = false>
__device__ __forceinline__ int lz4_decompress_wave(const gbyte* __restrict__ src_, const int32_t n_, gbyte* dst_, const int32_t cap_,
                                                   lbyte* lds_, const bool rap, const int32_t op0_ = 0)
{
    const gbyte* __restrict__ src = uni_ptr(src_);
    gbyte* dst = uni_ptr(dst_);
    lbyte* ring = uni_ptr(lds_);
    const int32_t n = (int32_t)uni((uint32_t)n_), cap = (int32_t)uni((uint32_t)cap_);
    const int32_t op0 = HIST ? (int32_t)uni((uint32_t)op0_) : 0;
    const int lane = lane_id();
    const uint32_t L = (uint32_t)lane;
    if (cap == 0) return (n == 1 && src[0] == 0) ? op0 : -1;  // lz4.c:3412-3416
    if (n <= 0) return -1;                                    // lz4.c:3418
    typedef __attribute__((address_space(3))) uint32_t L32;
    typedef u32x4 __attribute__((aligned(1))) U128;
    typedef __attribute__((address_space(1))) U128 G128;
    typedef uint32_t __attribute__((aligned(1))) U32;
    typedef __attribute__((address_space(3))) U32 S32;
    typedef uint16_t __attribute__((aligned(1))) U16;
    typedef __attribute__((address_space(3))) U16 S16;
    lbyte* recs = ring + kRing + 16;                          // 64 records of 4 bytes: offset | literal length << 16 | (match length - 4) << 24
    lbyte* cb = recs + kDecRecs * 4;                          // 64 bytes of history | the output block
    const int32_t iend = n, oend = op0 + cap;
    int32_t ip = 0, op = op0, rend = 0;                       // ring holds compressed bytes [.., rend)
    uint32_t nrec = 0, gbase = 0;                             // buffered sequences; where the first one's token is
    bool hist_ok = false;                                     // cb[0, 64) holds dst[op - 64, op)
    u32x4 pend = {0u, 0u, 0u, 0u}; bool have_pend = false;
#if defined(AOCLGPU_EXPERIMENTS) && defined(LZ4D_PROF)
    uint64_t dt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dprev = __builtin_amdgcn_s_memtime(); uint32_t dc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define DSTAGE(k) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); dt[k] += t_ - dprev; dprev = t_; } while (0)
#define DCOUNT(k, v) do { dc[k] += (v); } while (0)
#else
#define DSTAGE(k) do { } while (0)
#define DCOUNT(k, v) do { } while (0)
#endif
    for (;;) {
        // ---- keep the scan's 352 staged bytes ahead of ip (requested one iteration early; stored at once when the scan could not run) ----
        if (ip > rend) { rend = ip & ~15; have_pend = false; }
        if (have_pend) {
            *(__attribute__((address_space(3))) u32x4*)(ring + ((uint32_t)(rend + 16 * lane) & kRingMask)) = pend;
            if (((uint32_t)(rend + 16 * lane) & kRingMask) == 0) *(__attribute__((address_space(3))) u32x4*)(ring + kRing) = pend;   // mirror: 16-byte reads never wrap
            wave_fence();
            rend += 1024; have_pend = false;
        }
        if (rend - ip < 640 && rend < iend) {
            // (round 5: the chunk's last KiB is staged as well - zeros behind its end - and the scan stops at the tokens that reach
            //  the end instead of leaving the whole tail, ~120 sequences of a 256 KiB chunk, to the one-sequence path)
            const int32_t a_ = rend + 16 * lane;
            if (a_ + 16 <= iend) pend = *(const G128*)(src + a_);
            else {
                uint32_t w_[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int t_ = 0; t_ < 16; ++t_) if (a_ + t_ < iend) w_[t_ >> 2] |= (uint32_t)src[a_ + t_] << (8 * (t_ & 3));
                pend = u32x4{w_[0], w_[1], w_[2], w_[3]};
            }
            have_pend = true;
            if (rend - ip < 352) {
                *(__attribute__((address_space(3))) u32x4*)(ring + ((uint32_t)(rend + 16 * lane) & kRingMask)) = pend;
                if (((uint32_t)(rend + 16 * lane) & kRingMask) == 0) *(__attribute__((address_space(3))) u32x4*)(ring + kRing) = pend;
                wave_fence();
                rend += 1024; have_pend = false;
            }
        }
        // ---- scan: 64 bytes of tokens per round, at most one length-extension byte per field ----
        DSTAGE(0);
        int why = nrec > kDecRecs - 22 ? 0 : 3;               // 0 ran through its bytes (or the buffer has no room for another round: records
                                                              // that did not fit the last block), 2 a token the scan does not take, 3 input not staged
        while (nrec <= kDecRecs - 22 && rend - ip >= 352) {
            const uint32_t p = (uint32_t)ip + L;
            const uint32_t w1 = *(const S16*)(ring + (p & kRingMask));
            const uint32_t t = w1 & 0xFFu, e1 = w1 >> 8;
            uint32_t lit = t >> 4;
            const bool x1 = lit == 15;
            bool hard = x1 && e1 > 240;                       // 255: more bytes follow; above 240: the record's 8 bits do not hold the length
            if (x1) lit += e1;
            const uint32_t q = p + 1 + (x1 ? 1u : 0u) + lit;
            const bool x2 = (t & 15) == 15;
            hard = hard || q + 2 + (x2 ? 1u : 0u) + 6u > (uint32_t)iend;      // (a sequence that is not the block's last leaves 6 bytes behind it:
                                                                             //  what reaches the end is the one-sequence path's, with the reference's end rules)
            const uint32_t nx = hard ? 0x80000000u : q + 2 + (x2 ? 1u : 0u) - (uint32_t)ip;   // the next token's lane
            DSTAGE(1); DCOUNT(0, 1);
            // walk the real token starts: mask |= 1 << cur; cur = nx[cur] until cur >= 64 or a token the scan does not take
            uint64_t mask; uint32_t cur, stop;
            lz_walk_tokens(nx, mask, cur, stop);
            if (mask) {                                       // (a round adds at most 64 / 3 records)
                // Offset and match-length byte are read by the lanes that ARE tokens - a round has 8 or 9 of them - not by all 64:
                // where the next token starts does not depend on them, and an LDS instruction costs by its active lanes.  The
                // same lanes write their records at once.  (A match length past its extension byte is seen only now: the round
                // ends in front of that token, as the walk would have - its record and those behind it are not counted.)
                if (nrec == 0) gbase = (uint32_t)ip;
                const uint32_t idx = nrec + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                bool hard2 = false;
                if (__builtin_amdgcn_inverse_ballot_w64(mask)) {
                    const uint32_t w2 = *(const S32*)(ring + (q & kRingMask));
                    const uint32_t e2 = (w2 >> 16) & 0xFFu;
                    const uint32_t mcode = (t & 15) + (x2 ? e2 : 0u);  // match length - 4
                    hard2 = x2 && e2 > 240;
                    *(L32*)(recs + 4u * idx) = (w2 & 0xFFFFu) | (lit << 16) | (mcode << 24);
                }
                const uint64_t bad = ballot(hard2);
                if (__builtin_expect(bad != 0, 0)) { const int k = (int)__builtin_ctzll(bad); mask &= (1ull << k) - 1; cur = (uint32_t)k; stop = 2; }
                nrec += (uint32_t)__builtin_popcountll(mask); ip += (int32_t)cur;
                wave_fence();
            }
            why = (int)stop;
            DSTAGE(2); DCOUNT(1, (uint32_t)__builtin_popcountll(mask));
            if (why != 0 || rend - ip < 640) break;
        }
        if (why == 0 && nrec <= kDecRecs - 22) continue;
        // ---- execute buffered sequences, one per lane, until the buffer has room again (all of them in front of a sequential step) ----
        bool forced = false;
        while (nrec && !forced) {
            bool act = L < nrec;
            uint32_t lit = 0, mlen = 0, off = 0;
            if (act) { const uint32_t r = *(const L32*)(recs + 4u * L); off = r & 0xFFFFu; lit = (r >> 16) & 0xFFu; mlen = (r >> 24) + kMinMatch; }
            // where each sequence's token is (prefix sum of the compressed sizes) and where its output goes
            const uint32_t csz = act ? 3 + lit + (lit >= 15 ? 1u : 0u) + (mlen >= 19 ? 1u : 0u) : 0u;
            const uint32_t cinc = wave_scan_incl(csz);
            const uint32_t tokpos = gbase + cinc - csz, litsrc = tokpos + 1 + (lit >= 15 ? 1u : 0u);
            const uint32_t olen = lit + mlen;
            const uint32_t oinc = wave_scan_incl(olen);
            const uint32_t lpos = oinc - olen, mst = lpos + lit;          // block-relative: literals, match
            // what fits the block and the output buffer (with the reference's margin for its fast loop, lz4.c:3654); the
            // first sequence whose offset is 0 or reaches before the start of the output ends the group in front of it and
            // the sequential path decides what the reference decides (lz4.c:3745-3753, :3710)
            const int32_t room_o = oend - 32 - op;
            const uint32_t capo = room_o <= 0 ? 0u : ((uint32_t)room_o < kDecBlock ? (uint32_t)room_o : kDecBlock);
            uint32_t ng = (uint32_t)__builtin_popcountll(ballot(act && oinc <= capo));
            {
                // (also: a long match that starts further in front of the block than the history kept there)
                const int32_t srel_ = (int32_t)mst - (int32_t)off;
                const uint64_t badm = ballot(act && (off == 0 || off > (uint32_t)op + mst || (srel_ + (int32_t)mlen > 0 && srel_ < -(int32_t)kDecHist)));
                const uint32_t fb = badm ? (uint32_t)__builtin_ctzll(badm) : 64u;
                if (fb < ng) { ng = fb; forced = true; }
                else if (ng == 0) forced = true;              // the output buffer ends: the sequential path takes the next sequence
                if (forced) {
                    ip = (int32_t)lane_get(tokpos, (int)ng);             // back to that sequence's token; later records are dropped
                    // the scan had run ahead: a refill in flight would overwrite the oldest KiB of the ring, which holds ip again
                    if (rend - ip > 1024) have_pend = false;
                    if (rend - ip > (int32_t)kRing) rend = ip & ~15;     // further back than the ring holds: stage again from there
                }
                act = L < ng;
            }
            const uint32_t total = ng ? lane_get(oinc, (int)ng - 1) : 0u;
            if (ng) {
                DSTAGE(3); DCOUNT(2, 1); DCOUNT(3, ng);
                op += (int32_t)lz_group_execute<false>(src, dst, cb, hist_ok, (uint32_t)op, ng, act, lit, mlen, off, litsrc, lpos, mst, total);
                DSTAGE(6);
            }
            if (forced) nrec = 0;
            else {                                            // records that did not fit the block move to the front
                const uint32_t left = nrec - ng;
                for (uint32_t b0 = 0; b0 < left; b0 += kWave) {           // (front to back: a pass writes below what the next one reads)
                    uint32_t r = 0;
                    if (b0 + L < left) r = *(const L32*)(recs + 4u * (ng + b0 + L));
                    wave_fence();
                    if (b0 + L < left) *(L32*)(recs + 4u * (b0 + L)) = r;
                    wave_fence();
                }
                gbase += lane_get(cinc, (int)ng - 1);
                nrec = left;
                if (why == 0) break;                          // the buffer has room again: scan on
            }
        }
        if (why == 0 && !forced) continue;
        {
            const int r = (RAP && rap) ? lz4_decode_one_rap(src, iend, dst, oend, ip, op, lane) : lz4_decode_one(src, iend, dst, oend, ip, op, lane);
            if (r < 0) return r;
            if (r == 1) break;
            hist_ok = false;
            DSTAGE(7); DCOUNT(5, 1);
        }
    }
#if defined(AOCLGPU_EXPERIMENTS) && defined(LZ4D_PROF)
    if (lane == 0) { for (int k = 0; k < 8; ++k) { atomicAdd(&g_lz4_prof[k], (unsigned long long)dt[k]); atomicAdd(&g_lz4_prof[8 + k], (unsigned long long)dc[k]); } }
#endif
#undef DSTAGE
#undef DCOUNT
    return op;
}

}  // namespace

// MODE 0/1/2: every chunk of the launch uses that table mode (uniform batches); MODE 3: per-chunk choice (explicit
// batches with mixed sizes; 16 KB of LDS).  One kernel per mode keeps the register budget of the hot modes small.
template <int MODE, bool PROF>
__global__ void __launch_bounds__(64) lz4_compress_kernel(aoclgpu_batch b, uint32_t first)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[(MODE == 1 ? 9216 + LZ4_LDS_PAD : kTableBytes) + kSrcRing + kRecCap * rec_bytes(MODE)];
    const uint32_t c = first + blockIdx.x;
    if (c >= b.nchunks) return;
#ifdef AOCLGPU_EXPERIMENTS
    const uint64_t t_begin = __builtin_amdgcn_s_memtime();
#endif
    const ChunkView v = compress_view(b, c);
    const uint32_t n = v.src_len;
    int r;
    uint32_t lastrun = 0;
    if (n > kMaxInput) r = 0;                                               // lz4.c:2417
    else if (n == 0) {                                                      // lz4.c:2418-2429
        r = 0;
        if (v.dst_cap >= 1) { if (threadIdx.x == 0) v.dst[0] = 0; r = 1; }
    }
    else if (MODE == 3) {
        if (n < k64KLimit) r = lz4_compress_wave<0, false>(v.src, n, v.dst, v.dst_cap, (lbyte*)lds, b.reserved, &lastrun);
        else if (n <= (1u << 18) && !((b.reserved >> 15) != 0 && n > (1u << 17))) r = lz4_compress_wave<1, false>(v.src, n, v.dst, v.dst_cap, (lbyte*)lds, b.reserved, &lastrun);   // (accelerations: see mode_for)
        else r = lz4_compress_wave<2, false>(v.src, n, v.dst, v.dst_cap, (lbyte*)lds, b.reserved, &lastrun);
    }
    else if ((MODE == 0) != (n < k64KLimit) || (MODE == 1 && n > (1u << 18))) r = 0;   // launcher picked the wrong mode
    else r = lz4_compress_wave<MODE == 3 ? 2 : MODE, PROF>(v.src, n, v.dst, v.dst_cap, (lbyte*)lds, b.reserved, &lastrun);
#ifdef AOCLGPU_EXPERIMENTS
    if ((b.reserved & 2u) && b.aux) {                      // experiment build: per-chunk duration (s_memtime ticks / 1024) instead of the run length
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        lastrun = (uint32_t)((__builtin_amdgcn_s_memtime() - t_begin) >> 10);
    }
#endif
    if (threadIdx.x == 0) { b.out_len[c] = r; if (b.aux) b.aux[c] = lastrun; }
}

// The blocks of ONE stream with linked blocks (LZ4 frame format, blockMode LZ4F_blockLinked): explicit batch, decoded in
// order by a single wavefront into b.dst back to back; every block may copy from everything decoded before it (offsets are
// 16-bit, so from the previous 64 KB).  src_len[i] bit 31 marks a stored (uncompressed) block.  out_len[i] = decoded size
// of block i or < 0 (then every later block is < 0 as well); dst_cap[i] bounds block i, dst_off is ignored.
__global__ void __launch_bounds__(64) lz4_decompress_linked_kernel(aoclgpu_batch b, uint64_t dst_total)
{
    __shared__ __attribute__((aligned(16))) uint8_t ring[kDecLds];
    const int lane = lane_id();
    int32_t op = 0; bool bad = false;
    for (uint32_t c = 0; c < b.nchunks; ++c) {
        int32_t r = -1;
        if (!bad) {
            const gbyte* src = (const gbyte*)b.src + b.src_off[c];
            const uint32_t sl = b.src_len[c], n = sl & 0x7FFFFFFFu;
            uint32_t cap = b.dst_cap[c];
            if ((uint64_t)op + cap > dst_total) cap = (uint32_t)(dst_total - (uint64_t)op);
            if (sl >> 31) {
                if (n <= cap) { wave_copy((gbyte*)b.dst + op, src, n, lane); r = op + (int32_t)n; }
            } else r = lz4_decompress_wave<false, true>(src, (int32_t)n, (gbyte*)b.dst, (int32_t)cap, (lbyte*)ring, false, op);
            wave_fence();
        }
        if (r < 0) { bad = true; if (lane == 0) b.out_len[c] = -1; }
        else { if (lane == 0) b.out_len[c] = r - op; op = r; }
    }
}

// ------------------------------------------------------------------------------------------------
// Linked blocks on the compress side: the blocks of ONE LZ4 frame written with LZ4F_blockLinked, what LZ4F_compressFrame
// makes by default (lz4frame.c:373-419 -> LZ4F_compressBlock_continue :777-782 -> LZ4_compress_fast_continue lz4.c:3179-3256
// in prefix mode -> LZ4_compress_generic_validated :972-1360 with byU32, withPrefix64k, noDictIssue, limitedOutput).  The hash
// table carries over from block to block - indices are offsets from the frame's start, a match may begin up to 65535 bytes
// back in earlier blocks, the catch-up stops at the frame's start - so the blocks of a frame are one wavefront's work, in
// order.  Uniform batch: chunk c is block c, its output goes to slot c with room for (block size - 1) bytes as
// LZ4F_makeBlock asks (:747-749); out_len[c] = 0 when it does not fit (the caller stores the block) - the parse of such a
// block stopped where the output ran out and the table keeps what was written until then (lz4.c:1139), as in the reference.
// The parser is the chunk kernels' (lz4_compress_wave<2, ., LINKED>: batches of 64 consecutive positions resolved sequence
// after sequence, records, one-sequence-per-lane formatter) on stream indices; since round 4 - before that a one-match-per-batch
// loop of its own, 14 MiB/s for a 4 MiB frame.
// ------------------------------------------------------------------------------------------------
// One block of a stream through the chunk kernels' parser (lz4_compress_wave<2, ., LINKED>): `lds` = the stream's u32 table
// (16 KB, kept between blocks) followed by the source ring and the record buffer.  Returns the compressed size, 0 when it does
// not fit olimit bytes.
constexpr uint32_t kLinkLds = kTableBytes + kSrcRing + kRecCap * 16;
__device__ __forceinline__ uint32_t lz4_stream_block(const gbyte* __restrict__ src, const uint32_t b0, const uint32_t iend, lbyte* lds, const uint32_t accel,
                                                     gbyte* __restrict__ d, const uint32_t olimit, const uint32_t idxBase = 0, const uint32_t lowCand = 0,
                                                     const uint32_t lowPos = 0, const bool ext = false)
{
    const Lz4Link lk = {b0, idxBase, lowCand > idxBase ? lowCand : idxBase, lowPos, ext ? 1u : 0u};   // (what lies below the window is further than 64 KB away)
    uint32_t last = 0;
    const int r = lz4_compress_wave<2, false, true>(src, iend, d, olimit, lds, (accel - 1u) << 15, &last, &lk);
    return r > 0 ? (uint32_t)r : 0u;
}

__global__ void __launch_bounds__(64) lz4_compress_linked_kernel(aoclgpu_batch b)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[kLinkLds];
    typedef __attribute__((address_space(3))) uint32_t L32;
    const int lane = lane_id();
    for (uint32_t i = (uint32_t)lane; i < 4096; i += kWave) ((L32*)(lbyte*)lds)[i] = 0;  // LZ4_initStream
    wave_fence();
    const uint32_t accel = (b.reserved >> 15) + 1;
    for (uint32_t c = 0; c < b.nchunks; ++c) {
        const uint64_t b0 = (uint64_t)c * b.chunk_bytes;
        const uint64_t rest = b.total_bytes - b0;
        const uint32_t bn = (uint32_t)(rest < b.chunk_bytes ? rest : b.chunk_bytes);
        uint32_t r = 0;
        if (bn >= 1) {
            const uint64_t room = b.slot_bytes < (uint64_t)bn - 1 ? b.slot_bytes : (uint64_t)bn - 1;
            r = lz4_stream_block((const gbyte*)b.src, (uint32_t)b0, (uint32_t)b0 + bn, (lbyte*)lds, accel, (gbyte*)b.dst + (uint64_t)c * b.slot_bytes, (uint32_t)room);
        }
        wave_fence();
        if (lane == 0) b.out_len[c] = (int32_t)r;
    }
}

// The blocks one LZ4F_compressUpdate / LZ4F_flush call completes in a frame with linked blocks (csrc/lz4frame_api.cpp): one
// wavefront, block after block, on a window of the stream (the 64 KB before the call's first block + the call's blocks).  The
// stream's hash table lives in gtable between calls.
__global__ void __launch_bounds__(64) lz4_compress_stream_kernel(const uint8_t* src, uint32_t idxBase, const aoclgpu::Lz4StreamJob* jobs, uint32_t njobs,
                                                                 uint32_t* gtable, uint8_t* dst, uint32_t slot, int32_t* out_len, uint32_t accel)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[kLinkLds];
    typedef __attribute__((address_space(3))) uint32_t L32;
    L32* table = (L32*)(lbyte*)lds;
    const int lane = lane_id();
    for (uint32_t i = (uint32_t)lane; i < 4096; i += kWave) table[i] = gtable[i];
    wave_fence();
    for (uint32_t j = 0; j < njobs; ++j) {
        const uint32_t off = uni(jobs[j].off), len = uni(jobs[j].len), lowCand = uni(jobs[j].lowCand), lowPos = uni(jobs[j].lowPos), ext = uni(jobs[j].ext);
        uint32_t r = 0;
        if (len >= 1) {
            const uint32_t room = slot < len - 1 ? slot : len - 1;
            r = lz4_stream_block((const gbyte*)src, off, off + len, (lbyte*)lds, accel, (gbyte*)dst + (uint64_t)j * slot, room, idxBase, lowCand, lowPos, ext != 0);
        }
        wave_fence();
        if (lane == 0) out_len[j] = (int32_t)r;
    }
    wave_fence();
    for (uint32_t i = (uint32_t)lane; i < 4096; i += kWave) gtable[i] = table[i];
}

#ifndef LZ4D_WAVES
#define LZ4D_WAVES 6
#endif
template <bool RAP>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LZ4D_WAVES, LZ4D_WAVES))) lz4_decompress_kernel(aoclgpu_batch b)
{
    __shared__ __attribute__((aligned(16))) uint8_t ring[kDecLds];
    const uint32_t c = blockIdx.x;
    if (c >= b.nchunks) return;
#ifdef AOCLGPU_EXPERIMENTS
    const uint64_t t_begin = __builtin_amdgcn_s_memtime();
#endif
    const ChunkView v = decompress_view(b, c);
    int r;
    if (v.src_len > 0x7FFFFFFFu || v.dst_cap > 0x7FFFFFFFu) r = -1;
    else r = lz4_decompress_wave<RAP>(v.src, (int32_t)v.src_len, v.dst, (int32_t)v.dst_cap, (lbyte*)ring, RAP && c + 1 < b.nchunks);
    if (threadIdx.x == 0) b.out_len[c] = r;
#ifdef AOCLGPU_EXPERIMENTS
    if ((b.reserved & 2u) && b.aux) {                      // experiment build: per-chunk duration (s_memtime ticks / 64)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) b.aux[c] = (uint32_t)((__builtin_amdgcn_s_memtime() - t_begin) >> 6);
    }
#endif
}

}  // namespace aoclgpu

// ------------------------------------------------------------------------------------------------
// C-ABI launchers (include/aocl_gpu_codec.h)
// ------------------------------------------------------------------------------------------------
#include "host_util.h"

#ifdef AOCLGPU_EXPERIMENTS
extern "C" int aoclgpu_debug_lz4_cond(unsigned long long* host8)
{   /* experiment build: conditional-lane statistics since the last call (read and cleared) */
    if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(aoclgpu::g_lz4_cond), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[8] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(aoclgpu::g_lz4_cond), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
extern "C" int aoclgpu_debug_lz4_prof2(unsigned long long* host8)
{
    if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(aoclgpu::g_lz4_prof2), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(aoclgpu::g_lz4_prof2), z, sizeof(z));
    return 0;
}
extern "C" int aoclgpu_debug_lz4_prof(unsigned long long* host16)
{   /* experiment build: per-stage cycle sums accumulated by launches made with batch.reserved & 8 */
    if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(aoclgpu::g_lz4_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(aoclgpu::g_lz4_prof), z, sizeof(z));
    return 0;
}
#endif

extern "C" int aoclgpu_lz4_compress_chunks(const aoclgpu_batch* b, void* stream)
{
    if (int rc = aoclgpu::check_batch(b, /*decompress=*/false)) return rc;
    if (b->nchunks == 0) return AOCLGPU_OK;
    if (b->reserved & AOCLGPU_FLAG_LZ4_LINKED) {
        // the blocks of one LZ4 frame with linked blocks: one wavefront, block after block
        if (b->src_off || b->chunk_bytes < 65536 || b->total_bytes >= 0x7E000000ull) {
            aoclgpu::set_error_msg("lz4: linked blocks need a uniform batch of blocks of at least 64 KiB, below 0x7E000000 bytes in all");
            return AOCLGPU_ERR_ARG;
        }
        hipLaunchKernelGGL(aoclgpu::lz4_compress_linked_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, *b);
        return aoclgpu::check_launch("lz4_compress_linked_kernel");
    }
    // Table mode per launch: 0 = byU16 (16 KB LDS), 1 = packed byU32 (9 KB, chunks 65547 B .. 256 KiB), 2 = u32 with
    // check bits (16 KB).  Uniform batches launch the body and the ragged tail chunk separately; explicit batches
    // (sizes only known on the device) use the per-chunk dispatching kernel.
#ifdef AOCLGPU_EXPERIMENTS
    const bool prof = (b->reserved & 8u) != 0;
#else
    const bool prof = false; (void)prof;
#endif
    const bool forceU32 = (b->reserved & AOCLGPU_FLAG_LZ4_TABLE_U32) != 0;
    // (the packed table's 17-bit entries are swept for batches that span a few KB - acceleration 1; with LZ4_compress_fast's larger
    //  steps a batch of a chunk above 128 KiB may reach further than the sweeps allow: those take the u32 table)
    const bool accelerated = (b->reserved >> 15) != 0;
    auto mode_for = [&](uint64_t n) -> int { return n < 65547 ? 0 : ((n <= (1u << 18) && !forceU32 && !(accelerated && n > (1u << 17))) ? 1 : 2); };
    struct Launch { uint32_t first, count; int mode; } L[2]; int nl = 0;
    if (b->src_off) L[nl++] = { 0, b->nchunks, 3 };
    else {
        const uint64_t tail = b->total_bytes - (uint64_t)(b->nchunks - 1) * b->chunk_bytes;
        // a separate launch for the ragged tail chunk only when it needs another table mode (a lone chunk is slow)
        if (b->nchunks > 1 && mode_for(tail) != mode_for(b->chunk_bytes)) {
            L[nl++] = { b->nchunks - 1, 1, mode_for(tail) };
            L[nl++] = { 0, b->nchunks - 1, mode_for(b->chunk_bytes) };
        } else L[nl++] = { 0, b->nchunks, mode_for(b->nchunks > 1 ? b->chunk_bytes : tail) };
    }
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < nl; ++i) {
        const dim3 g(L[i].count), t(64);
        switch (L[i].mode) {
        case 0: hipLaunchKernelGGL((aoclgpu::lz4_compress_kernel<0, false>), g, t, 0, s, *b, L[i].first); break;
        case 1:
#ifdef AOCLGPU_EXPERIMENTS
            if (prof) { hipLaunchKernelGGL((aoclgpu::lz4_compress_kernel<1, true>), g, t, 0, s, *b, L[i].first); break; }
#endif
            hipLaunchKernelGGL((aoclgpu::lz4_compress_kernel<1, false>), g, t, 0, s, *b, L[i].first); break;
        case 2:
#ifdef AOCLGPU_EXPERIMENTS
            if (prof) { hipLaunchKernelGGL((aoclgpu::lz4_compress_kernel<2, true>), g, t, 0, s, *b, L[i].first); break; }
#endif
            hipLaunchKernelGGL((aoclgpu::lz4_compress_kernel<2, false>), g, t, 0, s, *b, L[i].first);
            break;
        default: hipLaunchKernelGGL((aoclgpu::lz4_compress_kernel<3, false>), g, t, 0, s, *b, L[i].first); break;
        }
        if (int rc = aoclgpu::check_launch("lz4_compress_kernel")) return rc;
    }
    return AOCLGPU_OK;
}

namespace aoclgpu {
int lz4_stream_blocks(const uint8_t* d_src, uint32_t idxBase, const Lz4StreamJob* d_jobs, uint32_t njobs, uint32_t* d_table, uint8_t* d_dst, uint32_t slot,
                      int32_t* d_len, uint32_t accel, void* stream)
{
    if (!njobs) return AOCLGPU_OK;
    hipLaunchKernelGGL(lz4_compress_stream_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_src, idxBase, d_jobs, njobs, d_table, d_dst, slot, d_len, accel);
    return check_launch("lz4_compress_stream_kernel");
}
// ------------------------------------------------------------------------------------------------
// The VALUE LZ4_decompress_safe returns for a malformed block: -(ip - src) - 1 with ip where the reference's decoder stood when
// it gave up (lz4.c:3799-3802).  The wave-parallel decoder above knows THAT a block is malformed, not where the reference's
// two-loop decoder notices it: AOCL_LZ4_decompress_generic (lz4.c:3367-3803) runs a fast loop while the output is more than 64
// bytes from its end and a safe loop afterwards, and the two check in a different order (a match length of 15+ with an offset
// outside the block fails BEFORE its extension bytes are read in the fast loop, behind them in the safe loop).  So a block the
// decoder rejected is walked once more here - tokens, lengths and offsets only, nothing is copied - along exactly the
// reference's control flow (endOnInput, full block, no dictionary), and the walk's result is the return value.  Cold path: one
// wavefront, scalar code; the stream is read through a 256-byte window held in one vector register.
//   litDist: distance from either buffer end at which the fast loop hands a long literal run to the safe loop - 64 in the
//   AVX build of the decoder (AOCL_LZ4_wildCopy64_AVX, :3448), which is what the reference runs on an AVX2 host once its
//   dispatcher has been set up; 32 in LZ4_decompress_generic (:4396).
__global__ void __launch_bounds__(64) lz4_error_value_kernel(const uint8_t* src_, const uint32_t n_, const uint32_t cap_, const uint32_t litDist_, int32_t* out)
{
    const gbyte* src = (const gbyte*)uni_ptr(src_);
    const int64_t iend = (int64_t)uni(n_), oend = (int64_t)uni(cap_), litDist = (int64_t)uni(litDist_);
    const int lane = lane_id();
    int64_t wbase = -1; uint32_t win = 0;
    auto B = [&](int64_t p) -> uint32_t {                     // the byte at src[p] (0 outside the block)
        if (p < 0 || p >= iend) return 0u;
        if (wbase < 0 || p < wbase || p >= wbase + 256) {
            wbase = p;
            const int64_t a = wbase + 4 * lane;
            win = 0;
            if (a + 4 <= iend) win = ld32(src + a);
            else for (int t = 0; t < 4; ++t) if (a + t < iend) win |= (uint32_t)src[a + t] << (8 * t);
        }
        const uint32_t o = (uint32_t)(p - wbase);
        return (lane_get(win, (int)(o >> 2)) >> (8 * (o & 3))) & 0xFFu;
    };
    int64_t ip = 0, op = 0, length = 0, offset = 0;
    uint32_t token = 0;
    int32_t result = 0;
    if (oend == 0) { result = (iend == 1 && B(0) == 0) ? 0 : -1; goto done; }          // :3412-3416
    if (iend == 0) { result = -1; goto done; }                                        // :3418
    if (oend - op >= 64) {
        for (;;) {                                            // ---- fast loop (:3427-3560) ----
            token = B(ip++); length = token >> 4;
            if (length == 15) {
                if (ip >= iend - 15) goto error;              // initial_error
                uint32_t s;
                do { s = B(ip); ip++; length += s; if (ip >= iend - 15) break; } while (s == 255);   // (loop_error is not an error here, :3440)
                if (op + length > oend - litDist || ip + length > iend - litDist) goto safe_literal_copy;
                ip += length; op += length;
            } else {
                if (ip > iend - 17) goto safe_literal_copy;
                ip += length; op += length;
            }
            offset = (int64_t)(B(ip) | (B(ip + 1) << 8)); ip += 2;
            length = token & 15;
            if (length == 15) {
                if (offset > op) goto error;                  // :3483: in front of the extension bytes
                uint32_t s;
                do { s = B(ip); ip++; length += s; if (ip >= iend - 4) goto error; } while (s == 255);
                length += 4;
                if (op + length >= oend - 64) goto safe_match_copy;
            } else {
                length += 4;
                if (op + length >= oend - 64) goto safe_match_copy;
                if (offset <= op && offset >= 8) { op += length; continue; }
            }
            if (offset > op) goto error;                      // :3511
            op += length;
        }
    }
    for (;;) {                                                // ---- safe loop (:3566-3790) ----
        token = B(ip++); length = token >> 4;
        if (length != 15 && ip < iend - 16 && op <= oend - 32) {                      // the two-stage shortcut (:3580-3611)
            op += length; ip += length;
            length = token & 15; offset = (int64_t)(B(ip) | (B(ip + 1) << 8)); ip += 2;
            if (length != 15 && offset >= 8 && offset <= op) { op += length + 4; continue; }
            goto copy_match;
        }
        if (length == 15) {
            if (ip >= iend - 15) goto error;
            uint32_t s;
            do { s = B(ip); ip++; length += s; if (ip >= iend - 15) break; } while (s == 255);
        }
    safe_literal_copy:
        if (op + length > oend - 12 || ip + length > iend - 8) {                      // :3628-3676: the last literals, or an error
            if (ip + length != iend || op + length > oend) goto error;
            op += length;
            result = (int32_t)op; goto done;
        }
        ip += length; op += length;
        offset = (int64_t)(B(ip) | (B(ip + 1) << 8)); ip += 2;
        length = token & 15;
    copy_match:
        if (length == 15) {
            uint32_t s;
            do { s = B(ip); ip++; length += s; if (ip >= iend - 4) goto error; } while (s == 255);
        }
        length += 4;
    safe_match_copy:
        if (offset > op) goto error;                          // :3710
        if (op + length > oend - 5) goto error;               // :3776 (only looked at above oend - 12)
        op += length;
    }
error:
    result = (int32_t)(-ip - 1);
done:
    if (lane == 0) *out = result;
}
}  // namespace aoclgpu

extern "C" int aoclgpu_lz4_error_value(const void* src_dev, uint32_t n, uint32_t cap, uint32_t lit_dist, int32_t* out, void* stream)
{
    hipLaunchKernelGGL(aoclgpu::lz4_error_value_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const uint8_t*)src_dev, n, cap, lit_dist, out);
    return aoclgpu::check_launch("lz4_error_value_kernel");
}

extern "C" int aoclgpu_lz4_decompress_chunks(const aoclgpu_batch* b, void* stream)
{
    if (int rc = aoclgpu::check_batch(b, /*decompress=*/true)) return rc;
    if (b->nchunks == 0) return AOCLGPU_OK;
    if (b->reserved & AOCLGPU_FLAG_LZ4_LINKED) {
        if (!b->src_off) { aoclgpu::set_error_msg("lz4: linked blocks need an explicit batch"); return AOCLGPU_ERR_ARG; }
        hipLaunchKernelGGL(aoclgpu::lz4_decompress_linked_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, *b, (uint64_t)b->total_bytes);
        return aoclgpu::check_launch("lz4_decompress_linked_kernel");
    }
    if (b->reserved & AOCLGPU_FLAG_RAP_PARTITIONS)
        hipLaunchKernelGGL(aoclgpu::lz4_decompress_kernel<true>, dim3(b->nchunks), dim3(64), 0, (hipStream_t)stream, *b);
    else
        hipLaunchKernelGGL(aoclgpu::lz4_decompress_kernel<false>, dim3(b->nchunks), dim3(64), 0, (hipStream_t)stream, *b);
    return aoclgpu::check_launch("lz4_decompress_kernel");
}
