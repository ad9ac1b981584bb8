// csrc/snappy_kernels.hip -- Snappy raw format for gfx950 (MI355X).
//
// Bit-exact wave-parallel restatement of
//   compress   : snappy::RawCompress -> Compress(Source*,Sink*) -> AOCL_CompressFragment
//                /root/reference algos/snappy/snappy.cc:2494-2666, :1749-1823, :846-1046
//                (EmitLiteral :435-476, EmitCopy :478-568, HashBytes :152-158, table sizing :619-657)
//   decompress : snappy::RawUncompress -> DecompressAllTags / SnappyArrayWriter
//                algos/snappy/snappy.cc:2271-2390, :1466-1570, :2131-2209
// The speculative 64-probe batch is the one described in lz4_kernels.hip; what differs is the schedule
// (skip += skip >> 5, so probe j sits at base + skip_j - 32; the 16 unrolled probes of :906-934 are its
// first 16 steps), the u16 table of 256..16384 entries (32 KB of LDS), the ip-1 back-fill + immediate
// re-probe after every copy (:1016-1021), and the tag-byte output format.
#include "common.hip.h"
#include "lz_group.hip.h"
#include <type_traits>
#include <stdlib.h>

namespace aoclgpu {
namespace {

constexpr uint32_t kBlock = 65536, kMaxTable = 1u << 14, kMinTable = 1u << 8, kMargin = 15;
constexpr uint32_t kSchedLen = 272;
#ifndef SN_FIRST_WIDTH
#define SN_FIRST_WIDTH 64        // lanes of the batch that follows a copy: consecutive positions ip, ip+1, ... (34 = one search's own steps)
#endif
static_assert(SN_FIRST_WIDTH >= 34 && SN_FIRST_WIDTH <= 64, "the first batch holds at least the schedule's consecutive steps");
constexpr int kSearchLanes = 34;     // a search that follows a copy: the re-probe of ip and the 33 one-byte steps of the schedule
// measured (8 GiB, JSON/log records), first batch = the schedule's own lanes: 64 lanes 244.9 ms, 48 227.6, 40 223.0, 34 217.4

// skip_j of the recurrence skip += skip >> 5 starting at 32 (snappy.cc:903-944); position offset = skip_j - 32.
struct SkipTable {
    uint32_t v[kSchedLen];
    constexpr SkipTable() : v() {
        uint32_t s = 32;
        for (uint32_t j = 0; j < kSchedLen; ++j) { v[j] = s; s += s >> 5; }
    }
};
__device__ const SkipTable g_skip{};

__device__ __forceinline__ uint32_t sn_off(uint32_t j)
{   // offset of probe j from the first probe of the schedule
    if (j <= 32) return j;
    if (j <= 48) return 32 + 2 * (j - 32);
    if (j <= 59) return 64 + 3 * (j - 48);
    if (j <= 67) return 97 + 4 * (j - 59);
    return (j < kSchedLen ? g_skip.v[j] : 0x7FFFFFFFu) - 32;
}

__device__ __forceinline__ uint32_t sn_hash(uint32_t bytes, int shift) { return (bytes * 0x1e35a7bdu) >> shift; }

__device__ __forceinline__ int log2floor(uint32_t x) { return 31 - __builtin_clz(x); }

// EmitLiteral (snappy.cc:435-476): tag + optional length bytes + data.  Returns bytes written.
__device__ __forceinline__ uint32_t emit_literal(gbyte* op, const gbyte* lit, uint32_t len, int lane)
{
    const uint32_t n = len - 1;
    uint32_t hdr = 1;
    if (n < 60) { if (lane == 0) op[0] = (uint8_t)(n << 2); }
    else {
        const uint32_t count = (uint32_t)(log2floor(n) >> 3) + 1;
        if (lane == 0) op[0] = (uint8_t)((59 + count) << 2);
        if ((uint32_t)lane < count) op[1 + lane] = (uint8_t)(n >> (8 * lane));
        hdr += count;
    }
    wave_copy(op + hdr, lit, len, lane);
    return hdr + len;
}

// EmitCopy (snappy.cc:478-568).  Returns bytes written.
__device__ __forceinline__ uint32_t emit_copy(gbyte* op, uint32_t offset, uint32_t len, int lane)
{
    uint32_t w = 0;
    if (len >= 12) {
        const uint32_t n64 = len >= 68 ? (len - 4) / 64 : 0;
        for (uint32_t e = (uint32_t)lane; e < n64; e += kWave) {
            gbyte* p = op + 3 * e;
            p[0] = (uint8_t)(2 | (63 << 2)); p[1] = (uint8_t)offset; p[2] = (uint8_t)(offset >> 8);
        }
        w = 3 * n64; len -= 64 * n64;
        if (len > 64) {
            if (lane == 0) { op[w] = (uint8_t)(2 | (59 << 2)); op[w + 1] = (uint8_t)offset; op[w + 2] = (uint8_t)(offset >> 8); }
            w += 3; len -= 60;
        }
    }
    if (len < 12 && offset < 2048) {
        if (lane == 0) { op[w] = (uint8_t)(1 | ((len - 4) << 2) | ((offset >> 8) << 5)); op[w + 1] = (uint8_t)offset; }
        return w + 2;
    }
    if (lane == 0) { op[w] = (uint8_t)(2 | ((len - 1) << 2)); op[w + 1] = (uint8_t)offset; op[w + 2] = (uint8_t)(offset >> 8); }
    return w + 3;
}

// One <=64 KiB fragment (AOCL_CompressFragment).  Returns bytes written.
struct Win { u32x4 a, b; };                              // bytes [p, p+16) and [p+16, p+32)
// 32-byte window [p, p+32): the 4 bytes of the test and 24 (of 28) bytes of forward extension in the lane's own registers - two
// 16-byte loads instead of two 8-byte loads plus the neighbours' registers by `ds_bpermute` (round 3 took bytes +12 .. +27 from
// lanes +8 and +16, which only consecutive lanes could do).  far = the second half lies inside the fragment (p + 32 <= n).
__device__ __forceinline__ Win load_win(const gbyte* __restrict__ src, uint32_t p, bool far)
{
    typedef u32x4 __attribute__((aligned(1))) U;
    typedef const __attribute__((address_space(1))) U* GP;
    Win w;
    w.a = *(GP)(src + p);
    w.b = *(GP)(src + (far ? p + 16 : p));               // (no branch: a lane near the end reads its first half twice and never looks at b)
    return w;
}

// Same batch machinery as lz4_kernels.hip (windowed loads, in-register extension up to 28 bytes, several copies
// resolved per batch, next windows requested early); Snappy specifics: the schedule is consecutive only for its
// first 33 probes, the table has no spare bits for check bits, no catch-up, matches may run to the fragment end.
// GT = false: the table lives in LDS (32 KB: 5 fragments resident per CU).  GT = true: the table lives in global memory
// (one 32 KB slice of a workspace per resident workgroup, served by L2 / Infinity Cache) and LDS only holds a 2 KB scratch
// for the slot-reuse rounds (indexed by the low 10 hash bits: a false alarm only cuts a batch short) - the number of
// resident fragments is then bounded by registers (20 per CU) instead of LDS, which is what a latency-bound parser needs.
// Table accesses in global memory are relaxed agent-scope atomics (served by L2, never by a stale L1 line); a wave's
// vector memory operations reach L2 in order, so the next batch's reads see this batch's commits.
typedef __attribute__((address_space(3))) uint16_t lu16;
typedef __attribute__((address_space(1))) uint32_t gu32;
constexpr uint32_t kScratch = 1024;

// Round 4: the parser does not write the block itself.  Every copy it decides becomes an 8-byte record in LDS (literal length in
// front of it | copy length << 17 | offset << 34) and when the buffer may overflow in the next batch the wave formats all buffered
// copies at once, ONE PER LANE: prefix sums give each its literal source and its output position, the literal tag, the literals
// (four pieces per size class, lz_group.hip.h), and the copy elements (EmitCopy's 64-byte elements, the 60-byte one, the last one
// of 2 or 3 bytes) are computed per lane.  A batch adds at most 17 copies (64 consecutive positions, a copy covers at least 4).
#ifndef SN_REC_CAP
#define SN_REC_CAP 60
#endif
constexpr uint32_t kSnRecCap = SN_REC_CAP, kSnRecFlush = kSnRecCap - 17, kSnRecBytes = kSnRecCap * 8;

// Experiment build (-DAOCLGPU_EXPERIMENTS -DSN_COUNT): memory accesses of the parser by origin, counted per lane and summed per
// table kind (index 0-7 LDS-table waves, 8-15 HBM-table waves): batches, probe-window loads (two 16-byte loads each), table slot
// reads, candidate-window loads (two 16-byte loads each), table slot writes, table bytes filled, extension-loop rounds.
#if defined(AOCLGPU_EXPERIMENTS) && defined(SN_COUNT)
__device__ unsigned long long g_sn_count[16];
#define SNC(k, cond) do { if (cond) snc[k]++; } while (0)
#else
#define SNC(k, cond) do { } while (0)
#endif

template <bool GT, int TAG = 0>
__device__ __attribute__((noinline)) uint32_t sn_compress_fragment(const gbyte* __restrict__ in_, const uint32_t n_,
                                                                   gbyte* __restrict__ out_, uint16_t* table_, lu16* scr, lbyte* recs_, int lane)
{
    const gbyte* __restrict__ in = uni_ptr(in_);
    gbyte* __restrict__ out = uni_ptr(out_);
    lbyte* recs = uni_ptr(recs_);
    const uint32_t n = uni(n_);
    // GT: a slot is a u32 = position | check bits of the 4 bytes at that position << 16.  The check bits never change a
    // decision (a probe whose check bits differ cannot pass the 4-byte test); they spare the speculative lanes the fetch
    // of a candidate window that cannot match - most of the kernel's memory traffic.  An empty slot aliases position 0
    // as in the reference's zero-filled table, so the table is filled with entry(0).
    using TP = typename std::conditional<GT, gu32*, lu16*>::type;
    TP table = (TP)uni_ptr((typename std::conditional<GT, uint32_t*, uint16_t*>::type)table_);
    auto chk = [&](uint32_t v4) -> uint32_t { return GT ? ((v4 * 2654435761u) & 0xFFFF0000u) : 0u; };
    auto tget = [&](uint32_t h) -> uint32_t {
        if (GT) return __hip_atomic_load((uint32_t*)(table + h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return table[h];
    };
#if defined(AOCLGPU_EXPERIMENTS) && defined(SN_COUNT)
    uint32_t snc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    auto tput = [&](uint32_t h, uint32_t v, uint32_t bytes4) {          // bytes4 = the 4 bytes at position v
        SNC(4, true);
        if (GT) __hip_atomic_store((uint32_t*)(table + h), (v & 0xFFFFu) | chk(bytes4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else table[h] = (uint16_t)v;
    };
    uint32_t tsize = n > kMaxTable ? kMaxTable : (n < kMinTable ? kMinTable : (2u << log2floor(n - 1)));   // :619-632
    const int shift = 32 - log2floor(tsize);
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    SNC(5, lane == 0);
    if (GT) {
        const uint32_t fill = n >= 4 ? chk(ld32(in)) : 0u;
        for (uint32_t i = (uint32_t)lane; i < tsize / 4; i += kWave)
            ((__attribute__((address_space(1))) u4*)table)[i] = u4{fill, fill, fill, fill};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    } else {
        for (uint32_t i = (uint32_t)lane; i < tsize / 8; i += kWave)     // memset, snappy.cc:651-657
            ((__attribute__((address_space(3))) u4*)table)[i] = u4{0, 0, 0, 0};
    }
    wave_fence();

    uint32_t ip = 0, op = 0, anchor = 0;                                  // anchor = next_emit
    // buffered copies [0, nrec): they describe in[f_anchor, anchor) and go to out + op
    uint32_t nrec = 0, f_anchor = 0;
    typedef __attribute__((address_space(3))) uint64_t L64;
    typedef u32x4 __attribute__((aligned(1))) U128;
    typedef __attribute__((address_space(1))) U128 G128;
    auto rec_put = [&](uint32_t idx, uint32_t lit, uint32_t len, uint32_t off) {
        *(L64*)(recs + 8u * idx) = (uint64_t)lit | ((uint64_t)len << 17) | ((uint64_t)off << 34);
    };
    auto flush = [&]() {                                   // EmitLiteral :435-476 + EmitCopy :478-568 for every buffered copy
        const uint32_t L = (uint32_t)lane;
        const bool act = L < nrec;
        uint32_t lit = 0, len = 0, off = 0;
        if (act) { const uint64_t r = *(const L64*)(recs + 8u * L); lit = (uint32_t)r & 0x1FFFFu; len = (uint32_t)(r >> 17) & 0x1FFFFu; off = (uint32_t)(r >> 34); }
        const uint32_t hdr = lit == 0 ? 0u : (lit <= 60 ? 1u : (lit <= 256 ? 2u : 3u));
        const uint32_t n64 = len >= 68 ? (len - 4) / 64 : 0u;
        uint32_t rem = len - 64 * n64;
        const bool big60 = rem > 64;
        if (big60) rem -= 60;
        const bool two = rem < 12 && off < 2048;
        const uint32_t csz = 3 * n64 + (big60 ? 3u : 0u) + (two ? 2u : 3u);
        const uint32_t osz = act ? hdr + lit + csz : 0u, isz = act ? lit + len : 0u;
        const uint32_t oinc = wave_scan_incl(osz), iinc = wave_scan_incl(isz);
        const uint32_t tp = op + oinc - osz;              // the literal's tag (or the first copy element)
        const uint32_t sp = f_anchor + iinc - isz;        // the literal's bytes
        const uint32_t lp = tp + hdr, cp = lp + lit;      // literal bytes, copy elements
        const gbyte* sl = in + sp; gbyte* d = out + lp;
        const bool c16 = act && lit >= 16 && lit <= 64, c4 = act && lit >= 4 && lit < 16, c1 = act && lit >= 1 && lit < 4;
        const uint32_t e16 = lit - 16, p16a = e16 < 16 ? e16 : 16u, p16b = e16 < 32 ? e16 : 32u;
        const uint32_t e4 = lit - 4, p4a = e4 < 4 ? e4 : 4u, p4b = e4 < 8 ? e4 : 8u;
        u32x4 q0 = {0u, 0u, 0u, 0u}, q1 = q0, q2 = q0, q3 = q0;
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        uint8_t b0 = 0, b1 = 0, b2 = 0;
        if (c16) { q0 = *(const G128*)sl; q1 = *(const G128*)(sl + p16a); q2 = *(const G128*)(sl + p16b); q3 = *(const G128*)(sl + e16); }
        if (c4) { r0 = ld32(sl); r1 = ld32(sl + p4a); r2 = ld32(sl + p4b); r3 = ld32(sl + e4); }
        if (c1) { b0 = sl[0]; b1 = sl[lit >> 1]; b2 = sl[lit - 1]; }
        if (act) {
            if (lit) {
                const uint32_t nl = lit - 1;
                out[tp] = (uint8_t)(nl < 60 ? (nl << 2) : ((nl < 256 ? 60u : 61u) << 2));
                if (nl >= 60) out[tp + 1] = (uint8_t)nl;
                if (nl >= 256) out[tp + 2] = (uint8_t)(nl >> 8);
            }
            gbyte* c = out + cp + 3 * n64;
            if (big60) { c[0] = (uint8_t)(2 | (59 << 2)); st16(c + 1, off); c += 3; }
            if (two) st16(c, (1u | ((rem - 4) << 2) | ((off >> 8) << 5)) | ((off & 0xFFu) << 8));
            else { c[0] = (uint8_t)(2 | ((rem - 1) << 2)); st16(c + 1, off); }
        }
        asm volatile("" ::: "memory");                    // (the loads above stay above)
        if (c16) { *(G128*)d = q0; *(G128*)(d + p16a) = q1; *(G128*)(d + p16b) = q2; *(G128*)(d + e16) = q3; }
        if (c4) { st32(d, r0); st32(d + p4a, r1); st32(d + p4b, r2); st32(d + e4, r3); }
        if (c1) { d[0] = b0; d[lit >> 1] = b1; d[lit - 1] = b2; }
        // rare: copies of 68 bytes and more (their 64-byte elements), literal runs beyond 64 bytes - one after the other, whole wave
        for (uint64_t m = ballot(act && n64 > 0); m; m &= m - 1) {
            const int k = (int)__builtin_ctzll(m);
            const uint32_t o = lane_get(cp, k), c = lane_get(n64, k), of = lane_get(off, k);
            for (uint32_t e = L; e < c; e += kWave) { gbyte* q = out + o + 3 * e; q[0] = (uint8_t)(2 | (63 << 2)); q[1] = (uint8_t)of; q[2] = (uint8_t)(of >> 8); }
        }
        for (uint64_t m = ballot(act && lit > 64); m; m &= m - 1) {
            const int k = (int)__builtin_ctzll(m);
            wave_copy(out + lane_get(lp, k), in + lane_get(sp, k), lane_get(lit, k), lane);
        }
        op += lane_get(oinc, 63); f_anchor = anchor; nrec = 0;
        wave_fence();
    };
    if (n >= kMargin) {
        const uint32_t ip_limit = n - kMargin;
        bool extra = false, have_pre = false; Win pre = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}}; uint32_t pre_fill = 0;
        uint32_t i0 = 0;
        for (;;) {
            asm volatile("; batch begin" ::: "memory");
            // -------- one batch of 64 speculative probes (snappy.cc:903-974, 1016-1032) --------
            const uint32_t base = ip + 1;
            const int32_t j = (int32_t)(i0 + (uint32_t)lane) - (extra ? 1 : 0);
            // A batch that follows a copy holds CONSECUTIVE positions ip + lane: lanes [0, 34) are the search's own steps, and
            // the lanes behind them are there for the searches that follow the copies this batch resolves (each of them probes
            // 34 consecutive positions of its own) - the mask-based fast-forward below runs through ~4 copies per batch that
            // way.  Every other batch follows the schedule of the one search it belongs to.
            const bool consb = extra && i0 == 0;
            const uint32_t pos = consb ? ip + (uint32_t)lane : base + sn_off((uint32_t)j);
            const bool valid = consb ? (lane == 0 || (uint64_t)ip + (uint32_t)lane + 1 <= ip_limit)
                                     : (uint64_t)base + sn_off((uint32_t)(j + 1)) <= ip_limit;          // :946-950
            // The first batch of a search only probes the schedule's consecutive steps (lanes [0, linlim)): a search in
            // compressible data ends there, and the 30 lanes behind them (steps of 2-4 bytes) would each cost a window and a
            // table slot - 64 bytes from the memory system for the HBM-table waves - that nobody looks at.  A search that
            // does run past them continues with a full batch of the sparse steps.  `act` = probed in this batch.
            const int linlim = (i0 == 0) ? (extra ? SN_FIRST_WIDTH : 33) : 0;   // lanes [0, linlim) probe consecutive positions
            const int width = i0 == 0 ? (extra ? SN_FIRST_WIDTH : kSearchLanes) : 64;
            const bool act = valid && lane < width;
            const bool far = (uint64_t)pos + 32 <= n;                     // (a candidate lies below its probe: its second half is inside too)
            Win pw = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
            SNC(0, lane == 0);
            if (i0 == 0 && have_pre) pw = pre;
            else if (act) { pw = load_win(in, pos, far); SNC(1, true); }
            if (extra && i0 == 0) {                                       // back-fill ip-1 (:1016-1017)
                const uint32_t fb = have_pre ? pre_fill : ld32(in + ip - 1);
                const uint32_t h1 = sn_hash(fb, shift);
                if (lane == 0) tput(h1, ip - 1, fb);
                wave_fence();
            }
            have_pre = false;
            const uint32_t seq = pw.a.x;
            const uint32_t h = sn_hash(seq, shift);
            uint32_t ent = 0;
            if (act) ent = tget(h);
            SNC(2, act);
            uint32_t cand = ent & 0xFFFFu;
            const bool plausible = act && (!GT || (ent & 0xFFFF0000u) == chk(seq));   // only these can pass the 4-byte test
            Win cw = {{~seq, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
            if (plausible) cw = load_win(in, cand, far);                  // the whole candidate window travels in one round of loads
            SNC(3, plausible);
            wave_fence();
            uint32_t w = (uint32_t)lane;
            if (GT) w = lds_first_lane_per_slot((uint32_t)(uintptr_t)(scr + (h & (kScratch - 1))), (uint32_t)lane, act);
            else {
                bool pending = act;
                do {                                                      // exact per-slot minimum lane
                    if (pending) table[h] = (uint16_t)lane;
                    wave_fence();
                    if (act) w = table[h];
                    wave_fence();
                    pending = act && (uint32_t)lane < w;
                } while (ballot(pending) != 0);
                if (act) table[h] = (uint16_t)cand;                       // undo the lane ids
            }
            wave_fence();
            const int e = first_lane(ballot(!valid));
            // Lanes that share their slot with an earlier lane of the batch ("duplicates") have a candidate that depends on the
            // parse: the position of the latest earlier lane on the slot whose table write has happened, else the table entry.
            // (With the 10-bit scratch of the HBM-table waves `w` also flags lanes whose full hashes differ: those turn out to
            // be ordinary lanes when they are looked at.)
            // In a batch of consecutive positions a duplicate STOPS the chain walk below, and is decided when the walk arrives
            // (round 4; lz4_kernels.hip has the same rule): the lanes below it are then known to be probes, copies or back-fills.
            // Every other batch keeps the plain rule: a duplicate is still an ordinary lane when NEITHER candidate can match
            // - the table's fails the 4-byte test and no earlier lane on the slot holds its 4 bytes - and the batch is cut at
            // the first one that could.
            const uint64_t tabok = ballot(act && cw.a.x == seq);
#if defined(AOCLGPU_EXPERIMENTS) && defined(SN_EXP_NOCUT)   /* (timing only: no duplicate is looked at - wrong parse, same work per batch) */
            const uint64_t dups = 0;
#else
            const uint64_t dups = ballot(act && w != (uint32_t)lane);
#endif
            int f = 64;
            if (!consb) for (uint64_t d = dups; d; d &= d - 1) {
                const int j = (int)__builtin_ctzll(d);
                const uint64_t same = ballot(act && h == lane_get(h, j)) & ((1ull << j) - 1);
                if (!same) continue;
                if (((tabok >> j) & 1) || (ballot(seq == lane_get(seq, j)) & same) != 0) { f = j; break; }
            }
            const int lim_e = e < width ? e : width;      // what the batch probes
            int lim = f < lim_e ? f : lim_e;              // lanes [0, lim) are decided exactly as the sequential parse decides them
            uint64_t okmask = tabok & (lim >= 64 ? ~0ull : ((1ull << lim) - 1));
            // table writes of lanes [a, b) U extra: a slot named by several of them goes to the latest (their writes happen in lane order)
            auto commit_lanes = [&](uint64_t r) -> bool {
                for (uint64_t d = dups & r; d; d &= d - 1) {
                    const int j = (int)__builtin_ctzll(d);
                    r &= ~(ballot(act && h == lane_get(h, j)) & ((1ull << j) - 1));
                }
                return (r >> lane) & 1;
            };
            auto owns = [&](int a, int b) -> bool {
                return commit_lanes((b >= 64 ? ~0ull : ((1ull << b) - 1)) & ~((1ull << a) - 1));
            };
            const uint32_t pos0 = lane_get(pos, 0);
            // lanes [0, linlim) probe consecutive positions pos0 + lane (the schedule's first 33 steps are 1 byte)
            // in-lane forward extension (<= 24 bytes beyond the 4 tested), FindMatchLength snappy-internal.h:228-352
            uint32_t fw; bool mf;
            auto extend = [&](const Win& c, uint32_t& fw_, bool& mf_) {
                const uint64_t x1 = (uint64_t)(pw.a.y ^ c.a.y) | ((uint64_t)(pw.a.z ^ c.a.z) << 32);
                const uint64_t x2 = (uint64_t)(pw.a.w ^ c.a.w) | ((uint64_t)(pw.b.x ^ c.b.x) << 32);
                const uint64_t x3 = (uint64_t)(pw.b.y ^ c.b.y) | ((uint64_t)(pw.b.z ^ c.b.z) << 32);
                const uint32_t x4 = pw.b.w ^ c.b.w;
                const uint32_t avail = far ? 28u : 8u;
                fw_ = x1 ? ((uint32_t)__builtin_ctzll(x1) >> 3) : 8;
                if (fw_ == 8 && far) fw_ += x2 ? ((uint32_t)__builtin_ctzll(x2) >> 3) : 8;
                if (fw_ == 16 && far) fw_ += x3 ? ((uint32_t)__builtin_ctzll(x3) >> 3) : 8;
                if (fw_ == 24 && far) fw_ += x4 ? ((uint32_t)__builtin_ctz(x4) >> 3) : 4;
                const uint32_t maxfwd = n - (pos + 4);
                mf_ = fw_ == avail && maxfwd > avail;
                fw_ = fw_ < maxfwd ? fw_ : maxfwd;
            };
            extend(cw, fw, mf);

            asm volatile("; chain begin" ::: "memory");
            // -------- resolve as many copies as this batch holds --------
            int start = 0;
            bool done = false, leave = false;
            auto request_next = [&]() {                   // the windows of the next batch (a search that starts at ip), requested early
                const uint32_t npos = ip + (uint32_t)lane;
                const bool nvalid = (lane == 0 || (uint64_t)ip + (uint32_t)lane + 1 <= ip_limit) && lane < SN_FIRST_WIDTH;
                pre.a = u32x4{0u, 0u, 0u, 0u}; pre.b = pre.a;
                if (nvalid) pre = load_win(in, npos, (uint64_t)npos + 32 <= n);
                pre_fill = ld32(in + ip - 1);
                have_pre = true;
            };
            // Fast-forward (the LZ4 kernel's mask-based resolve, lz4_kernels.hip): while the searches are the consecutive lanes
            // of the batch - a search that follows a copy probes ip, ip+1, ... - every copy whose match fits the in-register
            // windows is found by one readlane along precomputed `next match` links, and the table commit, the literal tags,
            // the literal bytes and the copy elements of ALL of them are derived from lane masks and written at once.  What
            // does not fit (a longer match, a search that leaves the consecutive lanes, the end of the fragment) is left to
            // the one-copy-at-a-time loop below, which starts where the fast-forward stopped.
            if (consb) {
                const int hi2 = lim < linlim ? lim : linlim;
                const uint64_t below2 = hi2 >= 64 ? ~0ull : ((1ull << hi2) - 1);
                const uint64_t dupv = dups & below2;
                const uint64_t okv = (tabok | dups) & below2;                          // matching lanes and the lanes that stop the walk
                uint32_t endl = (uint32_t)lane + 4u + fw;                              // match end, in lanes
                // the 34th probe of a search is followed by a 2-byte step: it is only made if that step stays inside (:946-950)
                auto own_lanes = [&](uint32_t s0) -> uint32_t { return (uint64_t)pos0 + s0 + 35 > ip_limit ? kSearchLanes - 1 : kSearchLanes; };
                auto next_of = [&](uint32_t en) -> uint32_t {                          // where the walk goes after a copy that ends at lane en
                    const uint64_t t = en < 64 ? okv >> en : 0;
                    const uint32_t cz = t ? (uint32_t)__builtin_ctzll(t) : 64u;
                    uint32_t nx = cz < own_lanes(en) ? en + cz : 64u;                 // 64: no match among the lanes of the search that starts at en
                    if (en >= (uint32_t)hi2) nx = 65;                                 // 65: the next search leaves the consecutive lanes
                    if (pos0 + en >= ip_limit) nx = 66;                               // 66: end of the fragment's searches (:1006-1008)
                    return nx;
                };
                uint32_t nxt = next_of(endl);
                uint64_t marks = 0, hitm = 0, dleft = dupv;   // dleft: duplicates nobody has looked at
                uint32_t fr = 0, ss = 0;                   // every duplicate below fr is decided or lies inside a copy; ss = where the search in progress began
                uint32_t mw = okv ? (uint32_t)__builtin_ctzll(okv) : 64u;
                if (mw >= own_lanes(0)) mw = 64;
                // the links the scalar walk follows (lz_group.hip.h: one readlane per copy); negative: the walk stops in front of
                // this lane - a duplicate (decided here, then the walk goes on) or a copy that needs the longer compare
                uint32_t nxw = (mf || ((dupv >> lane) & 1)) ? 0x80000000u : nxt;
                while (mw < 64) {
                    {
                        uint64_t mk; uint32_t stop, at;
                        lz_walk_tokens(nxw, mk, at, stop, uni(mw));
                        // (the results of an asm statement with several outputs count as divergent: readfirstlane of a scalar is free)
                        mw = uni(at); stop = uni(stop); mk = ((uint64_t)uni((uint32_t)(mk >> 32)) << 32) | uni((uint32_t)mk);
                        if (mk) { marks |= mk; ss = lane_get(endl, 63 - (int)__builtin_clzll(mk)); fr = ss; }
                        if (!stop) break;                  // 64 / 65 / 66: the walk is over
                    }
                    if (!((dleft >> mw) & 1)) break;       // needs the longer compare: the loop below
                    // A duplicate: its candidate is the position of the LATEST earlier lane on its slot whose table write has
                    // happened - as a probe, or as the ip-1 back-fill of the copy that covers it (:1016-1017) - else the table's entry.
                    uint32_t il = 64;
                    for (uint64_t L = ballot(act && h == lane_get(h, (int)mw)) & ((1ull << mw) - 1); L; ) {
                        const uint32_t x = 63u - (uint32_t)__builtin_clzll(L);
                        const uint64_t lowm = marks & ((2ull << x) - 1);             // copies that start at or below it (x <= 62)
                        bool wr = true;
                        if (lowm) { const int mm = 63 - (int)__builtin_clzll(lowm); const uint32_t em = lane_get(endl, mm); wr = (uint32_t)mm == x || em <= x || em == x + 1; }
                        if (wr) { il = x; break; }
                        L &= ~(1ull << x);
                    }
                    const bool written = il < 64;
                    const bool hit = written ? lane_get(seq, (int)il) == lane_get(seq, (int)mw) : ((tabok >> mw) & 1) != 0;
                    fr = mw + 1; dleft &= ~(1ull << mw);
                    if (!hit) {                            // a probe that finds nothing: the search goes on with its remaining lanes
                        const uint32_t ol = ss + own_lanes(ss);
                        const uint64_t rest = okv & ~((2ull << mw) - 1) & (ol >= 64 ? ~0ull : ((1ull << ol) - 1));
                        mw = rest ? (uint32_t)__builtin_ctzll(rest) : 64u;
                        continue;
                    }
                    hitm |= 1ull << mw;
                    const bool me = (uint32_t)lane == mw;
                    if (written) {                         // that lane's probe window is the candidate window
                        Win c2 = cw;
                        const uint32_t c0 = lane_get(pw.a.x, (int)il), c1 = lane_get(pw.a.y, (int)il), c2_ = lane_get(pw.a.z, (int)il), c3 = lane_get(pw.a.w, (int)il);
                        const uint32_t c4 = lane_get(pw.b.x, (int)il), c5 = lane_get(pw.b.y, (int)il), c6 = lane_get(pw.b.z, (int)il), c7 = lane_get(pw.b.w, (int)il);
                        if (me) { c2.a = u32x4{c0, c1, c2_, c3}; c2.b = u32x4{c4, c5, c6, c7}; }
                        uint32_t fw2; bool mf2;
                        extend(c2, fw2, mf2);
                        if (me) { fw = fw2; mf = mf2; cand = pos0 + il; endl = (uint32_t)lane + 4u + fw2; }
                        const uint32_t en = lane_get(endl, (int)mw), nx = next_of(en);
                        if (me) nxt = nx;
                    }
                    if (me) nxw = mf ? 0x80000000u : nxt;  // taken: the walk goes on from this lane (or stops at it for the longer compare)
                }
                {   // what the loop below may look at: the lanes in front of the first duplicate nobody has decided
                    const uint64_t und = fr >= 64 ? 0ull : (dups & (~0ull << fr));
                    const int f2 = und ? (int)__builtin_ctzll(und) : 64;
                    lim = f2 < lim_e ? f2 : lim_e; f = f2;
                    okmask = ((tabok & ~dups) | hitm) & (lim >= 64 ? ~0ull : ((1ull << lim) - 1));
                }
                if (marks) {
                    const int lastm = 63 - (int)__builtin_clzll(marks);
                    const uint32_t cur = lane_get(endl, lastm);                       // first lane after the last copy (may lie behind the batch)
                    const bool marked = __builtin_amdgcn_inverse_ballot_w64(marks);
                    // end of the last copy that starts below this lane (0: none): a DPP prefix maximum over the copies' ends
                    const uint32_t qend = wave_shr1(wave_scan_max(marked ? endl : 0u));
                    const bool mb_ = qend != 0;
                    const bool covered = marked || (uint32_t)lane < qend;
                    const uint64_t cover = ballot(covered);
                    const bool cont = mw != 66 && cur < (uint32_t)(lim < linlim ? lim : linlim);   // the loop below goes on inside this batch
                    {   // table writes the reference performs: every probed lane, and lane end-1 behind every copy (:1016-1017) - for
                        // the last copy only if its successor search is resolved from this batch (else the next batch's back-fill)
                        const uint64_t belowCur = cur >= 64 ? ~0ull : ~(~0ull << cur);
                        uint64_t ends = (cover << 1) & (~cover | marks);
                        if (!cont && cur < 64) ends &= ~(1ull << cur);
                        const uint64_t commit = (belowCur & ~(cover & ~marks)) | ((ends >> 1) & belowCur);
                        if (commit_lanes(commit)) tput(h, pos, seq);
                    }
                    wave_fence();
                    // one record per copy, in order behind the buffered ones (a batch of consecutive positions starts at the
                    // end of a copy: no literals are carried in)
                    const uint32_t lit = marked ? (uint32_t)lane - (mb_ ? qend : 0u) : 0u;       // literals in front of this copy (< 34)
                    const uint32_t idx = nrec + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(marks >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)marks, 0u));
                    if (marked) rec_put(idx, lit, 4u + fw, pos - cand);
                    nrec += (uint32_t)__builtin_popcountll(marks);
                    ip = pos0 + cur; anchor = ip;
                    extra = true; i0 = 0;
                    if (mw == 66) done = true;
                    else if (cont) start = (int)cur;
                    else { request_next(); leave = true; }
                }
            }
            if (!done && !leave) for (;;) {
                // lanes usable by the search being resolved: its own steps, as far as the batch holds them
                // (in a batch of consecutive positions the 34th probe of a search needs room for the 2-byte step behind it)
                const bool trunc = consb && (uint64_t)pos0 + (uint32_t)start + 35 > ip_limit;
                const int own = start + kSearchLanes - (trunc ? 1 : 0), cl = lim < linlim ? lim : linlim;
                const int hi = start == 0 ? (i0 == 0 && lim > own ? own : lim) : (cl < own ? cl : own);
                const uint64_t rem = okmask & (start ? (~0ull << start) : ~0ull) & (hi >= 64 ? ~0ull : ((1ull << hi) - 1));
                const int m = first_lane(rem);
                if (m >= 64) {                                            // lanes [start, hi) were probed without a match
                    if (owns(start, hi)) tput(h, pos, seq);
                    wave_fence();
                    // next_ip > ip_limit: emit_remainder.  A chained search walks 1-byte steps whatever the batch's
                    // own schedule was, so its end test is on the position after lane hi-1, not on `e`.
                    const bool at_end = (trunc && hi == own) || (start == 0 ? (hi == e && e < 64 && e <= f)
                                                                                   : ((uint64_t)pos0 + (uint32_t)hi + 1 > ip_limit));
                    if (at_end) { done = true; break; }
                    if (start == 0) i0 += (uint32_t)hi;
                    else i0 = (uint32_t)(hi - start);
                    break;
                }
                if (owns(start, m + 1)) tput(h, pos, seq);                           // probes the reference executed
                wave_fence();
                const uint32_t m_pos = lane_get(pos, m), m_cand = lane_get(cand, m);
                uint32_t matched = 4 + lane_get(fw, m);
                const bool more_fwd = lane_get((uint32_t)mf, m) != 0;
                if (more_fwd) {
                    for (;;) {
                        const uint32_t o = matched + 4u * (uint32_t)lane;
                        const uint32_t pa = m_pos + o;
                        uint32_t x = 0xFFFFFFFFu;
                        if (pa + 4 <= n) x = ld32(in + pa) ^ ld32(in + m_cand + o);
                        else if (pa < n) {
                            x = 0;
                            for (uint32_t t = 0; t < n - pa; ++t) x |= (uint32_t)(in[pa + t] ^ in[m_cand + o + t]) << (8 * t);
                            x |= 0xFFFFFFFFu << (8 * (n - pa));
                        }
                        SNC(6, lane == 0);
                        const int l = first_lane(ballot(x != 0));
                        if (l == 64) { matched += 256; continue; }
                        matched += 4u * (uint32_t)l + ((uint32_t)__builtin_ctz(lane_get(x, l)) >> 3);
                        break;
                    }
                }
                // -------- literal [anchor, m_pos) (:979; none after an immediate re-probe hit) + copy: one record --------
                if (lane == 0) rec_put(nrec, m_pos - anchor, matched, m_pos - m_cand);
                nrec++;
                ip = m_pos + matched;
                anchor = ip;
                if (ip >= ip_limit) { done = true; break; }               // :1006-1008
                extra = true; i0 = 0;
                // the next search (back-fill ip-1, probe ip, ip+1, ...) can be resolved from this batch when its probes
                // are consecutive lanes of the batch whose slots nobody before them touched
                const uint32_t Ln = ip - pos0;
                const int hi2 = lim < linlim ? lim : linlim;
                if (!more_fwd && (int)Ln < hi2) {
                    if ((uint32_t)lane == Ln - 1) tput(h, pos, seq);                  // back-fill ip-1 (:1016-1017)
                    wave_fence();
                    start = (int)Ln;
                    continue;
                }
                request_next();                              // otherwise request the windows of the next batch now
                break;
            }
            asm volatile("; chain end" ::: "memory");
            if (nrec > kSnRecFlush || done) { wave_fence(); flush(); }
            if (done) break;
        }
    }
    if (nrec) { wave_fence(); flush(); }
    if (anchor < n) op += emit_literal(out + op, in + anchor, n - anchor, lane);     // emit_remainder (:1039-1043)
#if defined(AOCLGPU_EXPERIMENTS) && defined(SN_COUNT)
    snc[5] *= GT ? tsize * 4u : tsize * 2u;
    for (int k = 0; k < 8; ++k) {
        uint32_t v = snc[k];
        for (int d = 32; d; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d);
        if (lane == 0 && v) atomicAdd(&g_sn_count[(GT ? 8 : 0) + k], (unsigned long long)v);
    }
#endif
    return op;
}

template <bool GT, int TAG = 0>
__device__ int sn_compress_wave(const gbyte* __restrict__ src, const uint32_t n, gbyte* __restrict__ dst,
                                const uint32_t cap, uint16_t* table, lu16* scr, lbyte* recs)
{
    const int lane = lane_id();
    if ((uint64_t)cap < 32ull + n + n / 6) return 0;                      // api/codec.cpp:271-274
    uint32_t op = 0, v = n;                                               // Varint::Encode32
    while (v >= 128) { if (lane == 0) dst[op] = (uint8_t)(v | 128); op++; v >>= 7; }
    if (lane == 0) dst[op] = (uint8_t)v;
    op++;
    for (uint32_t pos = 0; pos < n; pos += kBlock) {                      // snappy.cc:1761-1816
        const uint32_t frag = n - pos < kBlock ? n - pos : kBlock;
        op += sn_compress_fragment<GT, TAG>(src + pos, frag, dst + op, table, scr, recs, lane);
        wave_fence();
    }
    return (int)op;
}

// RawUncompress: returns the decoded length, or -1 for any malformed stream (the reference returns false).
constexpr uint32_t kRing = 2048, kRingMask = kRing - 1;

// one element (literal or copy), all rules of DecompressAllTags / SnappyArrayWriter (:1466-1570, :2131-2209).
// returns 0 = continue, <0 = malformed
__device__ int sn_decode_one(const gbyte* __restrict__ src, const uint32_t n, gbyte* dst, const uint32_t ulen,
                             uint32_t& ip, uint32_t& op, int lane)
{
    const uint32_t c = uni(src[ip]);
    const uint32_t type = c & 3;
    const uint32_t taglen = type == 0 ? ((c >> 2) < 60 ? 1 : (c >> 2) - 58) : (type == 1 ? 2 : (type == 2 ? 3 : 5));
    if (ip + taglen > n) return -1;                                       // RefillTag :1598-1653
    if (type == 0) {
        uint32_t len = (c >> 2) + 1;
        if (len > 60) {
            const uint32_t nb = len - 60;
            uint32_t v = 0;
            for (uint32_t i = 0; i < nb; ++i) v |= uni(src[ip + 1 + i]) << (8 * i);
            if (v == 0xFFFFFFFFu) return -1;
            len = v + 1;
        }
        ip += taglen;
        if (len > n - ip || len > ulen - op) return -1;                   // :2149-2158
        wave_copy(dst + op, src + ip, len, lane);
        ip += len; op += len;
    } else {
        uint32_t len, offset;
        if (type == 1) { len = 4 + ((c >> 2) & 7); offset = ((c >> 5) << 8) | uni(src[ip + 1]); }
        else if (type == 2) { len = (c >> 2) + 1; offset = uni(ld16(src + ip + 1)); }
        else { len = (c >> 2) + 1; offset = uni(ld32(src + ip + 1)); }
        ip += taglen;
        if (offset == 0 || offset > op || len > ulen - op) return -1;     // AppendFromSelf :2189-2207
        wave_fence();
        gbyte* d = dst + op;
        const gbyte* s = d - offset;
        if ((uint32_t)lane < len) d[lane] = s[offset >= kWave ? (uint32_t)lane : (uint32_t)lane % offset];
        wave_fence();
        op += len;                                                         // copy elements are <= 64 bytes
    }
    return 0;
}

// RawUncompress: decoded length, or -1 for any malformed stream (the reference returns false).  Same structure as the LZ4
// decoder (csrc/lz_group.hip.h): elements far from the end of the input are SCANNED 64 byte positions at a time from an LDS
// ring (lane k = "an element starts at ip + k", a scalar walk over the real starts) into 8-byte records - one per element,
// a literal run or a copy - and up to 64 buffered elements are EXECUTED at once, one per lane, through a 2 KB output block
// in LDS; everything else (literals with 2 - 4 length bytes, the end of the input, every rule that is broken) goes through
// the one-element path, which implements all checks of DecompressAllTags / SnappyArrayWriter.
#ifndef SND_RECS
#define SND_RECS 96         // round 5: more records than lanes - a group runs with all 64 lanes busy
#endif
constexpr uint32_t kSnRecs = SND_RECS;
constexpr uint32_t kSnDecLds = kRing + 32 + kSnRecs * 8 + kGrpLds;

__device__ __forceinline__ int sn_decompress_wave(const gbyte* __restrict__ src_, const uint32_t n_, gbyte* dst_, const uint32_t cap_,
                                                  lbyte* lds_, const bool rap)
{
    const gbyte* __restrict__ src = uni_ptr(src_);
    gbyte* dst = uni_ptr(dst_);
    lbyte* ring = uni_ptr(lds_);
    const uint32_t n = uni(n_), cap = uni(cap_);
    const int lane = lane_id();
    const uint32_t L = (uint32_t)lane;
    typedef __attribute__((address_space(3))) uint64_t L64;
    typedef uint64_t __attribute__((aligned(1))) U64;
    typedef __attribute__((address_space(3))) U64 S64;
    typedef u32x4 __attribute__((aligned(1))) U128;
    typedef __attribute__((address_space(1))) U128 G128;
    typedef __attribute__((address_space(3))) u32x4 LV;
    lbyte* recs = ring + kRing + 32;                          // 64 records: offset | (length | type << 16) << 32
    lbyte* cb = recs + kSnRecs * 8;                           // 64 bytes of history | the output block
    uint32_t ulen = 0, shift = 0, ip = 0;
    if (rap) ulen = cap;                                                  // RAP partition: no prefix, exact size known (snappy.cc:2567-2654)
    else for (;;) {                                                       // ReadUncompressedLength :1202-1223
        if (shift >= 32 || ip >= n) return -1;
        const uint32_t c = uni(src[ip]); ip++;
        const uint32_t val = c & 0x7f;
        if (shift > 0 && ((val << shift) >> shift) != val) return -1;
        ulen |= val << shift;
        if (c < 128) break;
        shift += 7;
    }
    if (ulen > cap) return -1;                                            // api/codec.cpp:296-303
    uint32_t op = 0, rend = 0;
    uint32_t nrec = 0, gbase = 0;                             // buffered elements; where the first one's tag is
    bool hist_ok = false;
    u32x4 pend = {0u, 0u, 0u, 0u}; bool have_pend = false;
    auto ring_put = [&]() {
        const uint32_t o = (rend + 16u * L) & kRingMask;
        *(LV*)(ring + o) = pend;
        if (o < 32) *(LV*)(ring + kRing + o) = pend;          // mirror: 32-byte reads never wrap
        wave_fence();
        rend += 1024; have_pend = false;
    };
    for (;;) {                                                            // DecompressAllTags :1466-1570
        int why = nrec > kSnRecs - 32 ? 0 : 3;                // 0 ran through its bytes (or no room for another round), 2 an element the scan does not take, 3 input not staged
        if (ip < n && why == 3) {
            if (ip > rend) { rend = ip & ~15u; have_pend = false; }
            if (have_pend) ring_put();
            if ((int32_t)(rend - ip) < 640 && rend < n) {
                // (round 5: the fragment stream's last KiB is staged as well - zeros behind its end - and the scan stops at the first
                //  element that does not lie inside the input instead of leaving the whole tail to the one-element path)
                const uint32_t a_ = rend + 16u * L;
                if ((uint64_t)a_ + 16 <= n) pend = *(const G128*)(src + a_);
                else {
                    uint32_t w_[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int t_ = 0; t_ < 16; ++t_) if (a_ + (uint32_t)t_ < n) w_[t_ >> 2] |= (uint32_t)src[a_ + t_] << (8 * (t_ & 3));
                    pend = u32x4{w_[0], w_[1], w_[2], w_[3]};
                }
                have_pend = true;
                if ((int32_t)(rend - ip) < 352) ring_put();
            }
            // ---- scan: 64 byte positions per round ----
            while ((int32_t)(rend - ip) >= 352) {
                const uint32_t p = ip + L;
                const uint64_t w = *(const S64*)(ring + (p & kRingMask));
                const uint32_t t = (uint32_t)w & 0xFFu, type = t & 3u, b1 = ((uint32_t)w >> 8) & 0xFFu;
                uint32_t len, off = 0, adv;
                bool hard = false;
                if (type == 0) {
                    const uint32_t l6 = t >> 2;
                    len = l6 < 60 ? l6 + 1 : b1 + 1;
                    hard = l6 > 60;                           // 2 - 4 length bytes: the one-element path
                    adv = (l6 < 60 ? 1u : 2u) + len;
                } else if (type == 1) { len = 4 + ((t >> 2) & 7); off = ((t >> 5) << 8) | b1; adv = 2; }
                else if (type == 2) { len = (t >> 2) + 1; off = ((uint32_t)w >> 8) & 0xFFFFu; adv = 3; }
                else { len = (t >> 2) + 1; off = (uint32_t)(w >> 8); adv = 5; }
                hard = hard || (uint64_t)p + adv > n;          // (an element that does not lie inside the input: the one-element path reports it)
                const uint32_t nx = hard ? 0x80000000u : L + adv;
                uint64_t mask; uint32_t cur, stop;
                lz_walk_tokens(nx, mask, cur, stop);
                why = (int)stop;
                if (mask) {                                   // (a round adds at most 64 / 2 records)
                    if (nrec == 0) gbase = ip;
                    const uint32_t idx = nrec + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                    if (__builtin_amdgcn_inverse_ballot_w64(mask)) *(L64*)(recs + 8u * idx) = (uint64_t)off | ((uint64_t)(len | (type << 16)) << 32);
                    nrec += (uint32_t)__builtin_popcountll(mask); ip += cur;
                    wave_fence();
                }
                if (why != 0 || nrec > kSnRecs - 32 || (int32_t)(rend - ip) < 640) break;   // (nrec <= 32 on entry: at most 64 after a round)
            }
            if (why == 0 && nrec <= kSnRecs - 32) continue;
        }
        // ---- execute buffered elements, one per lane, until the buffer has room again (all of them in front of a one-element step) ----
        bool forced = false;
        while (nrec && !forced) {
            bool act = L < nrec;
            uint32_t off = 0, len = 0, type = 0;
            if (act) { const uint64_t r = *(const L64*)(recs + 8u * L); off = (uint32_t)r; const uint32_t h = (uint32_t)(r >> 32); len = h & 0xFFFFu; type = h >> 16; }
            const uint32_t taglen = type == 0 ? (len <= 60 ? 1u : 2u) : (type == 1 ? 2u : (type == 2 ? 3u : 5u));
            const uint32_t lit = type == 0 ? len : 0u, mlen = type == 0 ? 0u : len;
            const uint32_t csz = act ? taglen + lit : 0u;
            const uint32_t cinc = wave_scan_incl(csz);
            const uint32_t tokpos = gbase + cinc - csz, litsrc = tokpos + taglen;
            const uint32_t oinc = wave_scan_incl(act ? len : 0u);
            const uint32_t lpos = oinc - (act ? len : 0u), mst = lpos;      // block-relative (an element is a literal run OR a copy)
            // what fits the block and the output (len > ulen - op is an error the one-element path reports, :2149-2158, :2189-2207);
            // the first copy whose offset is 0 or reaches before the start of the output ends the group in front of it
            const uint32_t room_o = ulen - op;
            const uint32_t capo = room_o < kGrpBlock ? room_o : kGrpBlock;
            uint32_t ng = (uint32_t)__builtin_popcountll(ballot(act && oinc <= capo));
            {
                const uint64_t badm = ballot(act && mlen && (off == 0 || off > op + mst));
                const uint32_t fb = badm ? (uint32_t)__builtin_ctzll(badm) : 64u;
                if (fb < ng) { ng = fb; forced = true; }
                else if (ng == 0) forced = true;              // the output ends: the one-element path decides
                if (forced) {
                    ip = lane_get(tokpos, (int)ng);           // back to that element's tag; later records are dropped
                    // the scan had run ahead: a refill in flight would overwrite the oldest KiB of the ring, which holds ip again
                    if ((int32_t)(rend - ip) > 1024) have_pend = false;
                    if ((int32_t)(rend - ip) > (int32_t)kRing) rend = ip & ~15u;   // further back than the ring holds: stage again from there
                }
                act = L < ng;
            }
            if (ng) {
                const uint32_t total = lane_get(oinc, (int)ng - 1);
                op += lz_group_execute<true>(src, dst, cb, hist_ok, op, ng, act, lit, mlen, off, litsrc, lpos, mst, total);
            }
            if (forced) nrec = 0;
            else {                                            // records that did not fit the block move to the front
                const uint32_t left = nrec - ng;
                for (uint32_t b0 = 0; b0 < left; b0 += kWave) {           // (front to back: a pass writes below what the next one reads)
                    uint64_t r = 0;
                    if (b0 + L < left) r = *(const L64*)(recs + 8u * (ng + b0 + L));
                    wave_fence();
                    if (b0 + L < left) *(L64*)(recs + 8u * (b0 + L)) = r;
                    wave_fence();
                }
                gbase += lane_get(cinc, (int)ng - 1);
                nrec = left;
                if (why == 0) break;                          // the buffer has room again: scan on
            }
        }
        if (why == 0 && !forced) continue;
        if (ip >= n) break;
        {
            const int r = sn_decode_one(src, n, dst, ulen, ip, op, lane);
            if (r < 0) return r;
            hist_ok = false;
        }
    }
    return op == ulen ? (int)op : -1;                                     // CheckLength :2146
}

}  // namespace

__global__ void __launch_bounds__(64) snappy_compress_kernel(aoclgpu_batch b)
{
    __shared__ __attribute__((aligned(16))) uint16_t table[kMaxTable];
    __shared__ __attribute__((aligned(16))) uint8_t recs[kSnRecBytes];
    const uint32_t c = blockIdx.x;
    if (c >= b.nchunks) return;
    const ChunkView v = compress_view(b, c);
    const int r = sn_compress_wave<false>(v.src, v.src_len, v.dst, v.dst_cap, table, nullptr, (lbyte*)recs);
    if (threadIdx.x == 0) b.out_len[c] = r;
}

// persistent grid, hash tables in a global workspace (tables + blockIdx.x * kMaxTable)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) snappy_compress_gt_kernel(aoclgpu_batch b, uint16_t* tables)
{
    __shared__ __attribute__((aligned(16))) uint16_t scr[kScratch];
    __shared__ __attribute__((aligned(16))) uint8_t recs[kSnRecBytes];
    uint16_t* table = tables + (size_t)blockIdx.x * kMaxTable * 2;       // u32[kMaxTable] per workgroup
    for (uint32_t c = blockIdx.x; c < b.nchunks; c += gridDim.x) {
        const ChunkView v = compress_view(b, c);
        const int r = sn_compress_wave<true, 2>(v.src, v.src_len, v.dst, v.dst_cap, table, (lu16*)scr, (lbyte*)recs);   // (TAG 2: this kernel's register budget, see the mixed kernel)
        if (threadIdx.x == 0) b.out_len[c] = r;
        wave_fence();
    }
}

// Mixed residency: one workgroup = three independent wavefronts (no barrier anywhere): wave 0 keeps its hash table in this
// workgroup's LDS, waves 1-2 keep theirs in HBM (check-bit slots, 2 KB LDS scratch each).  37.4 KB of LDS per workgroup: 4
// workgroups per CU = 4 LDS-table waves (the fast ones, no table traffic) + 8 HBM-table waves.  Chunks are handed out through
// an atomic counter, so the faster waves simply take more of them.  The HBM-table waves share the memory system's rate of
// scattered slot reads: with the round-4 parser eight of them per CU are faster than twelve (138.4 against 143.8 ms per 8 GiB;
// four per CU 167 ms, sixteen 160.8 ms - tools/experiments/README.md).
#ifndef SN_MIX_HBM
#define SN_MIX_HBM 2
#endif
#ifndef SN_MIX_WAVES_PER_SIMD
#define SN_MIX_WAVES_PER_SIMD 3
#endif
// (the kernel's register budget reaches the two non-inlined fragments only because they are instantiations of
//  their own - TAG 1 - that nobody else calls: the backend hands a callee the loosest budget among its callers, and shared with
//  the single-wave kernels the HBM-table fragment took 132 VGPRs when the budget was 128)
__global__ void __launch_bounds__(64 * (1 + SN_MIX_HBM)) __attribute__((amdgpu_waves_per_eu(SN_MIX_WAVES_PER_SIMD, SN_MIX_WAVES_PER_SIMD))) snappy_compress_mix_kernel(aoclgpu_batch b, uint32_t* tables, uint32_t* counter)
{
    __shared__ __attribute__((aligned(16))) uint16_t table[kMaxTable];
    __shared__ __attribute__((aligned(16))) uint16_t scr[SN_MIX_HBM][kScratch];
    __shared__ __attribute__((aligned(16))) uint8_t recs[1 + SN_MIX_HBM][kSnRecBytes];   // (32 KB + 2 x 2 KB + 3 x 480 B = 37.4 KB: four workgroups per CU)
    const uint32_t w = threadIdx.x >> 6;
    uint32_t* gt = tables + ((size_t)blockIdx.x * SN_MIX_HBM + (w ? w - 1 : 0)) * kMaxTable;
    for (;;) {
        uint32_t c = 0;
        if ((threadIdx.x & 63) == 0) c = atomicAdd(counter, 1u);
        c = uni(c);
        if (c >= b.nchunks) break;
        const ChunkView v = compress_view(b, c);
        int r;
        if (w == 0) r = sn_compress_wave<false, 1>(v.src, v.src_len, v.dst, v.dst_cap, table, nullptr, (lbyte*)recs[0]);
        else r = sn_compress_wave<true, 1>(v.src, v.src_len, v.dst, v.dst_cap, (uint16_t*)gt, (lu16*)scr[w - 1], (lbyte*)recs[w]);
        if ((threadIdx.x & 63) == 0) b.out_len[c] = r;
        wave_fence();
    }
}

#ifndef SND_WAVES
#define SND_WAVES 6
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SND_WAVES, SND_WAVES))) snappy_decompress_kernel(aoclgpu_batch b)
{
    const uint32_t c = blockIdx.x;
    if (c >= b.nchunks) return;
    __shared__ __attribute__((aligned(16))) uint8_t ring[kSnDecLds];
    const ChunkView v = decompress_view(b, c);
    const int r = sn_decompress_wave(v.src, v.src_len, v.dst, v.dst_cap, (lbyte*)ring, (b.reserved & AOCLGPU_FLAG_RAP_PARTITIONS) != 0);
    if (threadIdx.x == 0) b.out_len[c] = r;
}

}  // namespace aoclgpu

#include "host_util.h"

extern "C" int aoclgpu_snappy_compress_chunks(const aoclgpu_batch* b, void* stream)
{
    if (int rc = aoclgpu::check_batch(b, false)) return rc;
    if (b->nchunks == 0) return AOCLGPU_OK;
    // Batches that LDS can hold at once (5 tables per CU) use the LDS-table kernel; larger ones the persistent
    // global-table kernel with 14 workgroups per CU (the measured optimum with check bits in the slots: 10 -> 21.7,
    // 12 -> 22.9, 14 -> 24.3, 16 -> 23.6, 20 -> 19.7 GiB/s; without check bits 12 -> 21.8; LDS kernel 14.0).
    // AOCLGPU_FLAG_SNAPPY_TABLES_LDS / _HBM / _MIXED pin the choice (tests); the largest batches take the mixed kernel below.
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { aoclgpu::set_error_msg("snappy: device query failed"); return AOCLGPU_ERR_HIP; }
    const uint32_t cus = (uint32_t)prop.multiProcessorCount;
    aoclgpu::WsGuard guard;                               // held until the kernel that uses the scratch is enqueued
    const bool forceLds = (b->reserved & AOCLGPU_FLAG_SNAPPY_TABLES_LDS) != 0, forceHbm = (b->reserved & AOCLGPU_FLAG_SNAPPY_TABLES_HBM) != 0;
    const bool forceMix = (b->reserved & AOCLGPU_FLAG_SNAPPY_TABLES_MIXED) != 0;
    if ((!forceLds && !forceHbm && b->nchunks > 16u * cus) || forceMix) {
        // large batches: the mixed kernel, 4 workgroups of 1 + SN_MIX_HBM waves per CU (see snappy_compress_mix_kernel)
        uint32_t grid = 4u * cus;
        if (grid > (b->nchunks + SN_MIX_HBM) / (1 + SN_MIX_HBM)) grid = (b->nchunks + SN_MIX_HBM) / (1 + SN_MIX_HBM);
#ifdef AOCLGPU_EXPERIMENTS
        if (const char* g = getenv("AOCLGPU_SNAPPY_MIX_PER_CU")) grid = (uint32_t)(atoi(g) > 0 ? atoi(g) : 4) * cus;   // tuning hook
#endif
        const size_t tbytes = (size_t)grid * SN_MIX_HBM * aoclgpu::kMaxTable * sizeof(uint32_t);
        uint8_t* ws = (uint8_t*)aoclgpu::get_workspace(guard, tbytes + 256, stream);
        if (!ws) { aoclgpu::set_error_msg("snappy: workspace allocation failed"); return AOCLGPU_ERR_HIP; }
        uint32_t* counter = (uint32_t*)(ws + tbytes);
        if (hipMemsetAsync(counter, 0, 4, (hipStream_t)stream) != hipSuccess) { aoclgpu::set_error_msg("snappy: hipMemsetAsync failed"); return AOCLGPU_ERR_HIP; }
        hipLaunchKernelGGL(aoclgpu::snappy_compress_mix_kernel, dim3(grid), dim3(64 * (1 + SN_MIX_HBM)), 0, (hipStream_t)stream, *b, (uint32_t*)ws, counter);
        return aoclgpu::check_launch("snappy_compress_mix_kernel");
    }
    if (!forceLds && (b->nchunks > 5u * (uint32_t)prop.multiProcessorCount || forceHbm)) {
        int perCU = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, (const void*)aoclgpu::snappy_compress_gt_kernel, 64, 0) != hipSuccess || perCU < 1) {
            aoclgpu::set_error_msg("snappy: occupancy query failed"); return AOCLGPU_ERR_HIP;
        }
        if (perCU > 14) perCU = 14;
#ifdef AOCLGPU_EXPERIMENTS
        if (const char* g = getenv("AOCLGPU_SNAPPY_GT_PER_CU")) perCU = atoi(g) > 0 ? atoi(g) : perCU;      // tuning hook
#endif
        uint32_t grid = (uint32_t)(perCU * prop.multiProcessorCount);
        if (grid > b->nchunks) grid = b->nchunks;
        void* tables = aoclgpu::get_workspace(guard, (size_t)grid * aoclgpu::kMaxTable * sizeof(uint32_t), stream);
        if (!tables) { aoclgpu::set_error_msg("snappy: workspace allocation failed"); return AOCLGPU_ERR_HIP; }
        hipLaunchKernelGGL(aoclgpu::snappy_compress_gt_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, *b, (uint16_t*)tables);
        return aoclgpu::check_launch("snappy_compress_gt_kernel");
    }
    hipLaunchKernelGGL(aoclgpu::snappy_compress_kernel, dim3(b->nchunks), dim3(64), 0, (hipStream_t)stream, *b);
    return aoclgpu::check_launch("snappy_compress_kernel");
}

#if defined(AOCLGPU_EXPERIMENTS) && defined(SN_COUNT)
extern "C" int aoclgpu_debug_snappy_counts(unsigned long long* host16)
{   /* experiment build: accesses by origin accumulated by every compress launch since the last call (read and cleared) */
    if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(aoclgpu::g_sn_count), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(aoclgpu::g_sn_count), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int aoclgpu_snappy_decompress_chunks(const aoclgpu_batch* b, void* stream)
{
    if (int rc = aoclgpu::check_batch(b, true)) return rc;
    if (b->nchunks == 0) return AOCLGPU_OK;
    hipLaunchKernelGGL(aoclgpu::snappy_decompress_kernel, dim3(b->nchunks), dim3(64), 0, (hipStream_t)stream, *b);
    return aoclgpu::check_launch("snappy_decompress_kernel");
}
