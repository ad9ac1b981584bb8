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
#include "zstd_chain.h"
#include "zstd_hufspec.h"
#include "lz_group.hip.h"
#include "host_util.h"
#include <mutex>
#include <type_traits>
#include <atomic>
#include <stdlib.h>
#include <string.h>

namespace aoclgpu {
namespace {
using namespace zs;

constexpr u32 kBlockMax = 128u << 10;
constexpr u32 kMaxSeq = kBlockMax / 4 + 8;
constexpr u32 kDecHufPark = kBlockMax + 64 + 8192, kDecFsePark = kDecHufPark + 4096;   // (see DecLds: tables of a frame's earlier blocks)
constexpr u32 kDecWs = kDecFsePark + 5120;      // decoder workspace per resident block: literal buffer + a 12-bit Huffman table + the parked tables
constexpr u32 kSchedLen = 960;
constexpr u32 kRingBlk = 240;                   // source ring of the chain batches: two 240-byte blocks + a 32-byte mirror
}
constexpr uint32_t kSrcRingBytes = 512;
constexpr uint32_t kDfScratchBytes = 4096;
#ifndef DF_WAVES
#define DF_WAVES 2          // double-fast: independent wavefronts per workgroup (they share the entropy stage's LDS, one at a time)
#endif
constexpr uint32_t kDfWaves = DF_WAVES;
// (3 wavefronts per workgroup change nothing: the kernel's 135 VGPRs allow 12 per CU.  A register budget of 4 waves per SIMD
//  is not reached: with the entropy stage inlined under amdgpu_waves_per_eu(4,4) / amdgpu_num_vgpr(128) the allocator
//  stops at 132 VGPRs with 47 spills.)
namespace {

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
// the same schedule for another step size (levels below -1: stepSize = targetLength + 1, zstd_fast.c:440): three arrays of
// kSchedLen words in the workspace, filled by zstd_fast_sched_kernel in front of the frame kernel; nullptr = step 2
__device__ __forceinline__ void fast_sched_at(const u32* __restrict__ sched, const u32 m, u32& am, u32& sg, u32& st)
{
    am = 0x40000000u; sg = 2; st = 2;
    if (sched == nullptr) {
        if (m <= 62) am = 2 * m;
        else if (m < kSchedLen) { am = g_sched.a[m]; sg = g_sched.sigma[m]; st = g_sched.st[m]; }
    } else if (m < kSchedLen) { am = sched[m]; sg = sched[kSchedLen + m]; st = sched[2 * kSchedLen + m]; }
}

// ---- per-workgroup workspace in global memory ----
struct Workspace {
    u8  lits[kBlockMax + 64];
    u32 seqOff[kMaxSeq];     // offBase
    u32 seqLL[kMaxSeq];      // literal length
    u32 seqML[kMaxSeq];      // match length
    u32 seqSrc[kMaxSeq];     // position of the sequence's literals in the chunk
    u32 codesPad[4];         // (the state chains read code words four at a time, downwards from a sequence)
    u32 codes[kMaxSeq];      // LL code | ML code << 8 | OF code << 16 (ZSTD_seqToCodes)
    u16 fseState[3][kMaxSeq + 24];   // state of the LL / ML / OF chain BEFORE encode step e (round 6: the chains run in segments)
    u8  pad[16];             // (keeps sizeof a multiple of 16)
    u8  tmp[2 * kBlockMax + 1024];
};
static_assert(sizeof(Workspace) % 16 == 0, "workspaces of neighbouring workgroups stay 16-byte aligned");
typedef ZGLB Workspace GWorkspace;

// ---- LDS of the entropy stage (overlays the hash table once the match finder is done) ----
struct EntropyLds {
    u32 hist[256];           // byte histogram of the literals; then their code as one word per symbol (value | length << 16)
    u32 cnt3[3][64];         // histograms of the LL / ML / OF codes
    HufCT huf;
    HufScratch hs;           // (its FseScratch also serves the three sequence tables)
    FseCT ctLL, ctOF, ctML;
    u32 tile[WavePacker::kTileWords];
    u32 misc[16];
    u8  llbits[64], mlbits[64];   // LL_bits / ML_bits by code
};
typedef ZLDS EntropyLds LEntropy;
static_assert(sizeof(EntropyLds) <= 17920, "the entropy stage's LDS overlays the hash table of a 128 KiB chunk (17 920 bytes)");

template <int MLS>
__device__ __forceinline__ u32 z_hash(u64 v, u32 hlog)
{   // ZSTD_hash4Ptr / ZSTD_hash5Ptr / ZSTD_hash6Ptr (zstd_compress_internal.h:836-875)
    if (MLS == 4) return ((u32)v * 2654435761u) >> (32 - hlog);
    if (MLS == 5) return (u32)(((v << 24) * 889523592379ull) >> (64 - hlog));
    if (MLS == 6) return (u32)(((v << 16) * 227718039650203ull) >> (64 - hlog));
    return (u32)(((v << 8) * 58295818150454627ull) >> (64 - hlog));       // ZSTD_hash7Ptr (:848-850)
}

// common prefix of src[a..end) and src[b..], 256 bytes per step (ZSTD_count)
__device__ u32 wave_count(const gbyte* __restrict__ src, u32 a, u32 b, u32 end, int lane)
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

#ifdef ZSTD_PROF
__device__ unsigned long long g_zstd_gprof[4];      // gather: runs above 64 bytes, of them beyond the round's first four, of them above 1024 bytes, rounds
#endif
// Literal buffer of a block: gathered after the parse (ZSTD_storeSeq's copies + ZSTD_storeLastLiterals), so the search
// loop never waits on a load->store copy.  128 sequences per round, two consecutive ones per lane (their records are one
// 8-byte load per array, requested a round ahead): prefix sum of the literal lengths; runs up to 64 bytes as pieces at 0, k, 2k
// and len - k, every access inside the run; the round's first four runs above 64 bytes by the whole wave, 16 bytes per lane
// (the last piece overlapping) - and EVERY load of the round before its first store: one trip to memory per 128 sequences
// (until round 6: two per 64, and a third for every run above 24 bytes, which was most of this function's time).
// Returns the size of the last literal run.
// anchor_ == 0xFFFFFFFF: the caller only knows the block's total literal size (litTotal) - the last run is what the sequences leave
// (the double-fast finder kernel does not gather: its 32 waves per CU live on 64 registers; the entropy kernel gathers).
__device__ u32 gather_literals(const gbyte* __restrict__ src_, const u32 n_, GWorkspace* ws_, const u32 nbSeq_, const u32 anchor__, int lane, u32* seqLits = nullptr,
                               const u32 litTotal_ = 0)
{
    // (where this is a call, the arguments arrive in vector registers: scalar bases again, or every one of the round's ~50 accesses
    //  carries a 64-bit address pair of its own - the function then needs 194 registers)
    const gbyte* __restrict__ src = uni_ptr(src_); GWorkspace* ws = uni_ptr(ws_);
    const u32 n = uni(n_), nbSeq = uni(nbSeq_), anchor_ = uni(anchor__), litTotal = uni(litTotal_);
    typedef u32x4 __attribute__((aligned(1))) U128;
    typedef u32 __attribute__((aligned(1))) U32;
    typedef u64 __attribute__((aligned(1))) U64;
    g8* lits = ws->lits;
    const g32* gLL = ws->seqLL; const g32* gSrc = ws->seqSrc;
    auto rec2 = [&](const g32* a, u32 q) -> u64 {                        // records q, q+1 (the arrays have room behind nbSeq)
        u64 v = 0;
        if (q < nbSeq) v = *(const ZGLB U64*)(a + q);
        return v;
    };
    u32 lpos = 0;
    u64 llp = rec2(gLL, 2u * (u32)lane), spp = rec2(gSrc, 2u * (u32)lane);
    for (u32 q0 = 0; q0 < nbSeq; q0 += 2 * kWave) {
        const u32 qa = q0 + 2u * (u32)lane;
        const u64 nll = rec2(gLL, qa + 2 * kWave), nsp = rec2(gSrc, qa + 2 * kWave);
        u32 ll[2] = {(u32)llp, (u32)(llp >> 32)};
        const u32 sp[2] = {(u32)spp, (u32)(spp >> 32)};
        if (qa >= nbSeq) ll[0] = 0;
        if (qa + 1 >= nbSeq) ll[1] = 0;
        const u32 mine = ll[0] + ll[1];
        const u32 inc = wave_scan_incl(mine);
        u32 dp[2];
        dp[0] = lpos + inc - mine; dp[1] = dp[0] + ll[0];
        // the first four long runs of the round (uniform parameters)
        u64 lm0 = ballot(ll[0] > 64), lm1 = ballot(ll[1] > 64);
#if defined(ZSTD_PROF) && !defined(ZSTD_PROF_TOP)
        if (lane == 0) { const unsigned long long nl = (unsigned long long)(__builtin_popcountll(lm0) + __builtin_popcountll(lm1));
                         atomicAdd(&g_zstd_gprof[0], nl); atomicAdd(&g_zstd_gprof[1], nl > 4 ? nl - 4 : 0ull);
                         atomicAdd(&g_zstd_gprof[2], (unsigned long long)(__builtin_popcountll(ballot(ll[0] > 1024)) + __builtin_popcountll(ballot(ll[1] > 1024)))); atomicAdd(&g_zstd_gprof[3], 1ull); }
#endif
        u32 Ln[4] = {0, 0, 0, 0}, Ls[4] = {0, 0, 0, 0}, Ld[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (lm0) { const int k = (int)__builtin_ctzll(lm0); lm0 &= lm0 - 1; Ln[j] = lane_get(ll[0], k); Ls[j] = lane_get(sp[0], k); Ld[j] = lane_get(dp[0], k); }
            else if (lm1) { const int k = (int)__builtin_ctzll(lm1); lm1 &= lm1 - 1; Ln[j] = lane_get(ll[1], k); Ls[j] = lane_get(sp[1], k); Ld[j] = lane_get(dp[1], k); }
        }
        u32x4 p0[2], p1[2], p2[2], p3[2], lv[4]; u32 r0[2], r1[2], r2[2], r3[2], b0[2], b1[2], b2[2], lo_[4]; bool la[4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const u32 l = ll[t]; const gbyte* s = src + sp[t];
            const bool c16 = l >= 16 && l <= 64, c4 = l >= 4 && l < 16, c1 = l >= 1 && l < 4;
            p0[t] = p1[t] = p2[t] = p3[t] = u32x4{0u, 0u, 0u, 0u}; r0[t] = r1[t] = r2[t] = r3[t] = b0[t] = b1[t] = b2[t] = 0;
            if (c16) p0[t] = *(const ZGLB U128*)s;
            if (c16 && l > 32) p1[t] = *(const ZGLB U128*)(s + 16);
            if (c16 && l > 48) p2[t] = *(const ZGLB U128*)(s + 32);
            if (c16 && l > 16) p3[t] = *(const ZGLB U128*)(s + (l - 16));
            if (c4) r0[t] = *(const ZGLB U32*)s;
            if (c4 && l > 8) r1[t] = *(const ZGLB U32*)(s + 4);
            if (c4 && l > 12) r2[t] = *(const ZGLB U32*)(s + 8);
            if (c4 && l > 4) r3[t] = *(const ZGLB U32*)(s + (l - 4));
            if (c1) { b0[t] = s[0]; b1[t] = s[l >> 1]; b2[t] = s[l - 1]; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32 nfull = Ln[j] >> 4;                                // (a run's first 1024 bytes; what is behind them follows below)
            lo_[j] = (u32)lane < nfull ? 16u * (u32)lane : Ln[j] - 16;
            la[j] = Ln[j] != 0 && ((u32)lane < nfull || ((u32)lane == nfull && (Ln[j] & 15u) != 0));
            lv[j] = u32x4{0u, 0u, 0u, 0u};
            if (la[j]) lv[j] = *(const ZGLB U128*)(src + Ls[j] + lo_[j]);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const u32 l = ll[t]; g8* d = lits + dp[t];
            const bool c16 = l >= 16 && l <= 64, c4 = l >= 4 && l < 16, c1 = l >= 1 && l < 4;
            if (c16) *(ZGLB U128*)d = p0[t];
            if (c16 && l > 32) *(ZGLB U128*)(d + 16) = p1[t];
            if (c16 && l > 48) *(ZGLB U128*)(d + 32) = p2[t];
            if (c16 && l > 16) *(ZGLB U128*)(d + (l - 16)) = p3[t];
            if (c4) *(ZGLB U32*)d = r0[t];
            if (c4 && l > 8) *(ZGLB U32*)(d + 4) = r1[t];
            if (c4 && l > 12) *(ZGLB U32*)(d + 8) = r2[t];
            if (c4 && l > 4) *(ZGLB U32*)(d + (l - 4)) = r3[t];
            if (c1) { d[0] = (u8)b0[t]; d[l >> 1] = (u8)b1[t]; d[l - 1] = (u8)b2[t]; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) if (la[j]) *(ZGLB U128*)(lits + Ld[j] + lo_[j]) = lv[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) if (Ln[j] > 1024) wave_copy(lits + Ld[j] + 1024, src + Ls[j] + 1024, Ln[j] - 1024, lane);
        for (; lm0; lm0 &= lm0 - 1) { const int k = (int)__builtin_ctzll(lm0); wave_copy(lits + lane_get(dp[0], k), src + lane_get(sp[0], k), lane_get(ll[0], k), lane); }
        for (; lm1; lm1 &= lm1 - 1) { const int k = (int)__builtin_ctzll(lm1); wave_copy(lits + lane_get(dp[1], k), src + lane_get(sp[1], k), lane_get(ll[1], k), lane); }
        lpos += lane_get(inc, 63);
        llp = nll; spp = nsp;
    }
    const u32 anchor = anchor_ != 0xFFFFFFFFu ? anchor_ : n - (litTotal - lpos);
    wave_copy(lits + lpos, src + anchor, n - anchor, lane);
    wave_fence();
    if (seqLits) *seqLits = lpos;                                       // literals of the sequences (the parse need not count them)
    return n - anchor;
}

// ------------------------------------------------------------------------------------------------
// match finder: fills ws->seq*, ws->lits; returns nbSeq, *litSizeOut
// ------------------------------------------------------------------------------------------------
// Table layout: u16 per slot (low 16 bits of index = position + 2) and, for chunks above 16 KiB, one extra bit per
// slot in a packed word array (indices stay below 2^17 because nothing within 14 bytes of the chunk end is ever
// inserted).  17 KB of LDS per chunk instead of 32 KB lets 9 instead of 5 chunks be resident per CU.
#ifdef ZSTD_TRACE
__device__ unsigned long long g_ztrace[8 * 4096];
__device__ unsigned int g_ztrace_n;
#endif
#ifdef ZSTD_PROF
__device__ unsigned long long g_zstd_prof[16];
#ifdef ZSTD_PROF_NOWAIT   /* stamps without draining the global loads: a stage is charged the waits the compiler placed in it */
#define ZSTAGE(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); zacc[k] += t_ - zprev; zprev = t_; } while (0)
#else
#define ZSTAGE(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); zacc[k] += t_ - zprev; zprev = t_; } while (0)
#endif
#else
#define ZSTAGE(k) do { } while (0)
#endif
// entropy-stage profile (experiment build, tools/zstd_entropy_prof.py): drained stamps, previous stamp kept in EntropyLds::misc
#ifdef ZSTD_PROF
__device__ unsigned long long g_zstd_eprof[16];
#define ESTAGE(L, k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        if (lane_id() == 0) { ZLDS unsigned long long* p_ = (ZLDS unsigned long long*)(L)->misc; if ((k) >= 0) atomicAdd(&g_zstd_eprof[(k) < 0 ? 0 : (k)], t_ - *p_); *p_ = t_; } } while (0)
#else
#define ESTAGE(L, k) do { } while (0)
#endif
// ---- the chain walk of zstd_chain.h as scalar assembly (same inputs, same result; the C++ text stays the specification: the CPU
//      lane model runs it, -DZSTD_WALK_CXX puts it back into the kernel).  The compiler turns the C++ loop into ~140 scalar
//      instructions per sequence - conditions kept as 64-bit lane masks, converted through v_cndmask / v_readfirstlane, two dozen
//      phi copies per iteration; written by hand a table match takes 75 and a repcode 62.  Temporaries are fixed registers
//      s[83:101] (the operand list of an asm statement ends at 30).  FRAME: the match offset travels in its own VGPR.
//      Exit word: how (0 ran out of lanes, 1 search continues at lim, 3 special, 4 chain ends with the sequence, 5 no event)
//      | table match << 3 | post << 4 | event lane << 8 | E << 16 | search start << 24.
#define ZW_TA  "s[84:85]"
#define ZW_TB  "s[86:87]"
#define ZW_OFF_CHUNK "s_and_b32 s98, s88, 0x1ffff\n\t"
#define ZW_OFF_FRAME "v_readlane_b32 s98, %[offv], s93\n\t"
#define ZW_SEQ_BLOCK \
    "s_mov_b32 m0, %[rl]\n\t" \
    "v_writelane_b32 %[qll], s97, m0\n\t" \
    "v_writelane_b32 %[qoff], s98, m0\n\t" \
    "v_writelane_b32 %[qml], s99, m0\n\t" \
    "s_add_i32 %[rl], %[rl], 1\n\t" \
    "s_mov_b32 %[arel], s94\n\t" \
    "s_mov_b32 %[S], s94\n\t" \
    "s_sub_i32 s88, s94, 2\n\t" \
    "s_cmp_ge_u32 s88, %[lim]\n\t" \
    "s_cbranch_scc1 94f\n\t"                            /* table writes behind the trustworthy lanes */ \
    "s_cmp_lg_u32 s101, 0\n\t" \
    "s_cbranch_scc1 94f\n\t"                            /* immediate rep2 test not decidable from the maps */ \
    "s_lshl_b64 " ZW_TA ", -1, s83\n\t" \
    "s_add_i32 s89, s93, 2\n\t" \
    "s_lshl_b64 " ZW_TB ", -2, s89\n\t" \
    "s_andn2_b64 " ZW_TA ", " ZW_TA ", " ZW_TB "\n\t" \
    "s_or_b64 %[commit], %[commit], " ZW_TA "\n\t" \
    "s_bitset1_b64 %[commit], s88\n\t" \
    "s_branch 1b\n"
#define ZSTD_WALK_ASM(OFFLINE) \
    "1:\n\t"                                            /* ---- loop top ---- */ \
    "s_cmp_ge_u32 %[S], %[lim]\n\t" \
    "s_cbranch_scc1 90f\n\t" \
    "s_lshr_b64 " ZW_TA ", %[q1], 2\n\t" \
    "s_lshr_b64 " ZW_TA ", " ZW_TA ", %[S]\n\t" \
    "s_and_b32 s84, s84, 0x55555555\n\t" \
    "s_and_b32 s85, s85, 0x55555555\n\t" \
    "s_lshr_b64 " ZW_TB ", %[ok], %[S]\n\t" \
    "s_ff1_i32_b64 s91, " ZW_TA "\n\t"                  /* dr, -1: none */ \
    "s_ff1_i32_b64 s92, " ZW_TB "\n\t"                  /* dm, -1: none */ \
    "s_mov_b32 s83, %[S]\n\t"                           /* Sev */ \
    "s_cmp_lt_u32 s92, s91\n\t" \
    "s_cbranch_scc0 30f\n\t"                            /* no table match in front of the first repcode event */ \
    /* ---- table match at lane j (the fall-through: the common event) ---- */ \
    "s_add_i32 s93, %[S], s92\n\t"                      /* j */ \
    "s_and_b32 s89, %[ends], 0xff\n\t"                  /* end1 */ \
    "s_add_i32 s88, s93, 5\n\t"                         /* no repcode event at all: the tests are known below end1 - 5 only */ \
    "s_cmp_eq_u32 s91, -1\n\t" \
    "s_cselect_b32 s88, s88, 0\n\t" \
    "s_cmp_ge_u32 s88, s89\n\t" \
    "s_cbranch_scc1 95f\n\t" \
    "v_readlane_b32 s88, %[info], s93\n\t" \
    "s_bfe_u32 s95, s88, 0x50011\n\t"                   /* fw */ \
    "s_bfe_u32 s96, s88, 0x30016\n\t"                   /* catch-up code */ \
    OFFLINE                                             /* s98 = offset */ \
    "s_sub_i32 s90, s93, %[arel]\n\t"                   /* room */ \
    "s_min_u32 s96, s96, s90\n\t"                       /* back */ \
    "s_add_i32 s94, s93, s95\n\t" \
    "s_add_i32 s94, s94, 4\n\t"                         /* E */ \
    "s_add_i32 s88, s94, 4\n\t" \
    "s_cmp_le_u32 s88, s89\n\t" \
    "s_cselect_b32 s100, 1, 0\n\t"                      /* imm */ \
    "s_lshr_b64 " ZW_TA ", %[q1], s94\n\t" \
    "s_and_b32 s88, s84, s100\n\t"                      /* immediate rep2 match (rep2 = the old rep1) */ \
    "s_lshr_b64 " ZW_TA ", %[hard], s93\n\t" \
    "s_and_b32 s84, s84, 1\n\t" \
    "s_or_b32 s88, s88, s84\n\t" \
    "s_xor_b32 s101, s100, 1\n\t"                       /* post */ \
    "s_sub_i32 s97, s93, s96\n\t" \
    "s_sub_i32 s97, s97, %[arel]\n\t"                   /* ll */ \
    "s_add_i32 s99, s96, s95\n\t" \
    "s_add_i32 s99, s99, 4\n\t"                         /* ml */ \
    "s_cmp_eq_u32 s96, 5\n\t" \
    "s_cbranch_scc1 93f\n\t"                            /* more than four bytes of catch-up possible */ \
    "s_cmp_lg_u32 s88, 0\n\t" \
    "s_cbranch_scc1 93f\n\t" \
    "s_mov_b32 %[rep2], %[rep1]\n\t" \
    "s_mov_b32 %[rep1], s98\n\t" \
    "s_add_i32 s98, s98, 3\n\t"                         /* offBase */ \
    "s_mov_b64 %[q2], %[q1]\n\t" \
    "v_readlane_b32 s84, %[eqb], s93\n\t" \
    "s_lshr_b32 s84, s84, 4\n\t" \
    "s_mov_b32 s85, 0\n\t" \
    "s_lshl_b64 %[eq1], " ZW_TA ", s93\n\t" \
    "s_lshr_b64 " ZW_TA ", %[eq1], 1\n\t" \
    "s_and_b64 " ZW_TA ", " ZW_TA ", %[eq1]\n\t" \
    "s_lshr_b64 %[q1], " ZW_TA ", 2\n\t" \
    "s_and_b64 %[q1], %[q1], " ZW_TA "\n\t" \
    "s_min_u32 s88, s93, 36\n\t" \
    "s_add_i32 s88, s88, 28\n\t" \
    "s_lshl_b32 s89, s89, 8\n\t" \
    "s_or_b32 %[ends], s89, s88\n" \
    /* ---- one sequence: park the record, move on ---- */ \
    ZW_SEQ_BLOCK \
    "30:\n\t" \
    "s_cmp_eq_u32 s91, -1\n\t" \
    "s_cbranch_scc1 95f\n\t"                            /* neither */ \
    /* ---- repcode found by even probe k at k + 2 ---- */ \
    "s_add_i32 s93, %[S], s91\n\t"                      /* k */ \
    "s_cmp_ge_u32 s93, %[lim]\n\t" \
    "s_cbranch_scc1 91f\n\t" \
    "s_add_i32 s88, s93, 1\n\t" \
    "s_lshr_b64 " ZW_TA ", %[eq1], s88\n\t"             /* e1k */ \
    "s_and_b32 s96, s84, 1\n\t"                         /* back1 */ \
    "s_lshr_b64 " ZW_TA ", " ZW_TA ", 5\n\t" \
    "s_not_b64 " ZW_TA ", " ZW_TA "\n\t" \
    "s_ff1_i32_b64 s95, " ZW_TA "\n\t"                  /* f */ \
    "s_add_i32 s94, s93, s95\n\t" \
    "s_add_i32 s94, s94, 6\n\t"                         /* E */ \
    "s_add_i32 s88, s94, 4\n\t" \
    "s_bfe_u32 s89, %[ends], 0x80008\n\t"               /* end2 */ \
    "s_min_u32 s90, %[rep2], 1\n\t"                     /* rep2 != 0 */ \
    "s_cmp_le_u32 s88, s89\n\t" \
    "s_cselect_b32 s100, s90, 0\n\t"                    /* imm */ \
    "s_lshr_b64 " ZW_TA ", %[q2], s94\n\t" \
    "s_and_b32 s88, s84, s100\n\t"                      /* immediate rep2 match */ \
    "s_and_b32 s89, %[ends], 0xff\n\t"                  /* end1 */ \
    "s_xor_b32 s101, s100, s90\n\t"                     /* post = rep2 != 0 && !imm */ \
    "s_sub_i32 s97, s93, s96\n\t" \
    "s_sub_i32 s97, s97, %[arel]\n\t" \
    "s_add_i32 s97, s97, 2\n\t"                         /* ll */ \
    "s_mov_b32 s98, 1\n\t"                              /* offBase */ \
    "s_add_i32 s99, s96, s95\n\t" \
    "s_add_i32 s99, s99, 4\n\t"                         /* ml */ \
    "s_cmp_ge_u32 s94, s89\n\t" \
    "s_cbranch_scc1 92f\n\t"                            /* the run leaves the known part of the map */ \
    "s_cmp_lg_u32 s88, 0\n\t" \
    "s_cbranch_scc1 92f\n\t" \
    ZW_SEQ_BLOCK                                        /* (its own copy of the sequence block: no taken branch to it) */ \
    /* ---- exits ---- */ \
    "90:\n\t" "s_mov_b32 %[xw], 0\n\t" "s_branch 99f\n" \
    "91:\n\t" "s_mov_b32 %[xw], 1\n\t" "s_branch 99f\n" \
    "95:\n\t" "s_mov_b32 %[xw], 5\n\t" "s_branch 99f\n" \
    "92:\n\t" "s_mov_b32 %[xw], 3\n\t" "s_branch 96f\n" \
    "93:\n\t" "s_mov_b32 %[xw], 11\n\t" "s_branch 96f\n" \
    "94:\n\t" "s_mov_b32 %[xw], 4\n" \
    "96:\n\t" \
    "s_lshl_b32 s101, s101, 4\n\t" \
    "s_lshl_b32 s93, s93, 8\n\t" \
    "s_lshl_b32 s94, s94, 16\n\t" \
    "s_lshl_b32 s83, s83, 24\n\t" \
    "s_or_b32 %[xw], %[xw], s101\n\t" \
    "s_or_b32 s93, s93, s94\n\t" \
    "s_or_b32 %[xw], %[xw], s83\n\t" \
    "s_or_b32 %[xw], %[xw], s93\n" \
    "99:"

template <bool FRAME>
__device__ __forceinline__ zchain::WalkOut walk_scalar(const u64 okmask, const u64 hard, const int lim_, int32_t arel_, u32 rep1, u32 rep2, u64 eq1, const u64 eq2,
                                                       const u32 infoV, const u32 eqbV, const u32 offV, u32& q_ll, u32& q_off, u32& q_ml, u32& nbSeq,
                                                       const u32 qbase, const int S0, const u64 commit0)
{
    auto uni64 = [](u64 v) -> u64 { return ((u64)uni((u32)(v >> 32)) << 32) | uni((u32)v); };
    eq1 = uni64(eq1); rep1 = uni(rep1); rep2 = uni(rep2);
    const u64 okm = uni64(okmask), hardm = uni64(hard);
    u64 q1 = zchain::quad(eq1), q2 = zchain::quad(uni64(eq2)), commit = uni64(commit0);
    u32 S = uni((u32)S0), arel = uni((u32)arel_), ends = 64u | (64u << 8), rl = uni(nbSeq - qbase), xw;
    const u32 lim = uni((u32)lim_);
    if constexpr (FRAME)
        asm volatile(ZSTD_WALK_ASM(ZW_OFF_FRAME)
            : [S] "+s"(S), [arel] "+s"(arel), [rep1] "+s"(rep1), [rep2] "+s"(rep2), [eq1] "+s"(eq1), [q1] "+s"(q1), [q2] "+s"(q2), [ends] "+s"(ends),
              [commit] "+s"(commit), [rl] "+s"(rl), [qll] "+v"(q_ll), [qoff] "+v"(q_off), [qml] "+v"(q_ml), [xw] "=&s"(xw)
            : [ok] "s"(okm), [hard] "s"(hardm), [lim] "s"(lim), [info] "v"(infoV), [eqb] "v"(eqbV), [offv] "v"(offV)
            : "scc", "m0", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "s100", "s101");
    else
        asm volatile(ZSTD_WALK_ASM(ZW_OFF_CHUNK)
            : [S] "+s"(S), [arel] "+s"(arel), [rep1] "+s"(rep1), [rep2] "+s"(rep2), [eq1] "+s"(eq1), [q1] "+s"(q1), [q2] "+s"(q2), [ends] "+s"(ends),
              [commit] "+s"(commit), [rl] "+s"(rl), [qll] "+v"(q_ll), [qoff] "+v"(q_off), [qml] "+v"(q_ml), [xw] "=&s"(xw)
            : [ok] "s"(okm), [hard] "s"(hardm), [lim] "s"(lim), [info] "v"(infoV), [eqb] "v"(eqbV)
            : "scc", "m0", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "s100", "s101");
    // (the outputs of an asm statement with several results count as divergent; readfirstlane of a scalar register costs
    //  nothing and gives what follows its scalar compares and branches back)
    xw = uni(xw); S = uni(S); arel = uni(arel); rep1 = uni(rep1); rep2 = uni(rep2); ends = uni(ends); rl = uni(rl); commit = uni64(commit);
    nbSeq = qbase + rl;
    // how the chain ended (zstd_chain.h, behind the loop)
    const u32 how = xw & 7u, el = (xw >> 8) & 0xFFu, E = (xw >> 16) & 0xFFu, Sev = xw >> 24, end1 = ends & 0xFFu;
    zchain::WalkOut o;
    o.outcome = zchain::ZC_RESTART; o.S = (int)S; o.R = 0; o.ev = 0; o.el = 0; o.why = 0; o.arel = (int32_t)arel; o.rep1 = rep1; o.rep2 = rep2;
    o.ntail = 0; o.tail[0] = o.tail[1] = o.tail[2] = 0; o.post = false;
    if (how == 1) { o.R = (int)lim; commit |= zchain::span(S, lim); o.outcome = zchain::ZC_CONTINUE; }
    else if (how == 5) {
        const u32 R = end1 - 5 < lim ? end1 - 5 : lim;
        if (R < lim && S > 0) o.outcome = end1 < 64 ? zchain::ZC_REFRESH : zchain::ZC_RESTART;
        else { o.R = (int)R; commit |= zchain::span(S, R); o.outcome = zchain::ZC_CONTINUE; }
    }
    else if (how == 3) { commit |= zchain::span(S, el + 1); o.outcome = zchain::ZC_SPECIAL; o.ev = (xw & 8u) ? 1 : 2; o.el = (int)el; }
    else if (how == 4) {
        const u32 w3 = E - 2;
        u64 bits = (3ull << el) << 1;                                   // lanes el + 1, el + 2 (bits 64, 65 fall off)
        if (w3 < 64u) bits |= 1ull << w3;
        commit |= zchain::span(Sev, el + 1) | (bits & (~0ull >> (64u - lim)));   // (1 <= lim <= 64)
        o.tail[0] = (int)el + 1; o.tail[1] = (int)el + 2; o.tail[2] = (int)w3;
        o.ntail = (int)((el + 1 >= lim ? 1u : 0u) + (el + 2 >= lim ? 1u : 0u) + (w3 >= lim ? 1u : 0u));
        o.post = (xw & 16u) != 0;
    }
    o.commit = commit;
    return o;
}

// window of a frame (ZSTD_window_t): indices are position + 2
struct FrameWindow { u32 dictLimit, lowLimit, maxDist; };
__device__ __forceinline__ u32 lowest_prefix(const FrameWindow& w, u32 curr)
{   // ZSTD_getLowestPrefixIndex without a dictionary
    return (curr - w.dictLimit > w.maxDist) ? curr - w.maxDist : w.dictLimit;
}
// FRAME = false: the only block of a fresh frame (a chunk): table cleared here, u16 slots (+ 1 bit), repeat offsets {1,4,8}.
// FRAME = true: block [bs, n) of a frame whose bytes start at src[0] (zstd_frame_wave): T = u32 slots holding position + 2 as
// offsets into the frame, the table persists, candidates below the prefix start (ZSTD_getLowestPrefixIndex) are ignored and
// the catch-up stops there, the repeat offsets come in and go out (zstd_fast.c:481-489, :614-634); the source ring counts its
// 240-byte blocks from bs.
template <int MLS, typename T, bool HI, bool FRAME = false>
__device__ u32 fast_block(const gbyte* __restrict__ src_, const u32 n_, const u32 hlog_, lbyte* lds_, Workspace* ws_,
                          u32* litSizeOut, int lane, const u32 expf_, const u32 bs_ = 0, const FrameWindow* win = nullptr,
                          u32* rep1io = nullptr, u32* rep2io = nullptr, const u32* sched_ = nullptr)
{
    const u32* __restrict__ sched = FRAME ? uni_ptr(sched_) : nullptr;   // (another step size than 2: frames only)
    static_assert(FRAME ? (sizeof(T) == 4 && !HI) : sizeof(T) == 2, "u16 slots for chunks, u32 slots for frames");
    // (arguments of a non-inlined function arrive in VGPRs and count as divergent: everything the control flow hangs on is
    //  made wave-uniform here, or the scalar walk below is compiled as exec-masked vector code - common.hip.h)
    const gbyte* __restrict__ src = uni_ptr(src_);
    lbyte* lds = uni_ptr(lds_); GWorkspace* ws = (GWorkspace*)uni_ptr(ws_);   // (typed: the record stores are global_*, not flat_*)
    const u32 n = uni(n_), hlog = uni(hlog_), expf = uni(expf_);
    const u32 bs = FRAME ? uni(bs_) : 0u;                               // (a compile-time zero for chunks)
    typedef __attribute__((address_space(3))) T LT;
    typedef __attribute__((address_space(3))) u32 L32;
    constexpr bool kHi = HI;                                            // positions need 17 bits (chunks above 16 KiB)
    const u32 tsize = 1u << hlog;
    LT* table = (LT*)lds;
    L32* hib = (L32*)(lds + sizeof(T) * tsize);                         // kHi: tsize/32 words
    if constexpr (!FRAME) {
        for (u32 i = (u32)lane; i < tsize / 2; i += kWave) ((L32*)lds)[i] = 0;
        if (kHi) for (u32 i = (u32)lane; i < tsize / 32; i += kWave) hib[i] = 0;
        wave_fence();
    }
    u32 prefixIdx = 2;                                                   // table entries below it are not candidates
    if constexpr (FRAME) prefixIdx = lowest_prefix(*win, n + 2);        // (:443-446)
    const u32 prefixPos = prefixIdx - 2;
    auto tget = [&](u32 h) -> u32 {
        u32 v = table[h];
        if (kHi) v |= ((hib[h >> 5] >> (h & 31)) & 1u) << 16;
        return v;
    };
    auto tput = [&](u32 h, u32 v) {
        table[h] = (T)v;
        if (kHi) {
            if (v >> 16) __hip_atomic_fetch_or(&hib[h >> 5], 1u << (h & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_and(&hib[h >> 5], ~(1u << (h & 31)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    // u32 entries: index (position + 2, 18 bits) plus 14 check bits of the 4 bytes at that position.  The check
    // bits never change a decision; they spare the speculative lanes a cache-line fetch for hopeless candidates.
    constexpr bool kChk = false;
    constexpr u32 kPosMask = 0x3FFFFu;
    auto chk = [&](u32 v4) -> u32 { return kChk ? ((v4 * 2654435761u) & ~kPosMask) : 0u; };
    // Sequence records are parked one per lane and written 64 at a time (coalesced stores): a store in the search loop
    // would sit in front of every following load in vmcnt order.  Three fields per record; where a sequence's literals
    // start (seqSrc, for gather_literals) follows from the sequences before it - they leave no gaps - and is a prefix
    // sum at flush time.  The chain walk parks up to 16 records per call without looking: room is made in front of it.
    u32 q_ll = 0, q_off = 0, q_ml = 0;
    u32 nbSeq = 0, qbase = 0, srcNext = bs;                              // record qbase sits in lane 0; its literals start at srcNext
    auto flush_seq = [&](u32 count) {
        const u32 adv = (u32)lane < count ? q_ll + q_ml : 0u;
        const u32 inc = wave_scan_incl(adv);
        if ((u32)lane < count) {
            ws->seqLL[qbase + lane] = q_ll; ws->seqOff[qbase + lane] = q_off; ws->seqML[qbase + lane] = q_ml; ws->seqSrc[qbase + lane] = srcNext + inc - adv;
        }
        srcNext += lane_get(inc, 63); qbase += count;
    };
    auto park_seq = [&](u32 ll, u32 off, u32 ml) {                      // (wave-uniform arguments: one v_writelane per field)
        const u32 l = uni(nbSeq - qbase);
        asm("s_mov_b32 m0, %6\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\tv_writelane_b32 %2, %5, m0"
            : "+v"(q_ll), "+v"(q_off), "+v"(q_ml) : "s"(uni(ll)), "s"(uni(off)), "s"(uni(ml)), "s"(l) : "m0");
        nbSeq++;
    };
    auto put_seq = [&](u32 ll, u32 off, u32 ml) {
        park_seq(ll, off, ml);
        if (nbSeq - qbase == 64u) flush_seq(64);
    };
    u32 anchor = bs, s = bs + (bs == prefixPos ? 1u : 0u), rep1 = 1, rep2 = 0;   // rep {1,4,8}: 4 > maxRep is parked (:484-489)
    u32 saved1 = 0, saved2 = 0;
#ifdef ZSTD_PREFETCH
    u32 pfacc = 0, pfv = 0;
#endif
    if constexpr (FRAME) {
        rep1 = *rep1io; rep2 = *rep2io;
        const u32 curr = s + 2, maxRep = curr - lowest_prefix(*win, curr);
        if (rep2 > maxRep) { saved2 = rep2; rep2 = 0; }
        if (rep1 > maxRep) { saved1 = rep1; rep1 = 0; }
    }
#ifdef ZSTD_PROF
    unsigned long long zacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, zprev = __builtin_amdgcn_s_memtime(), zbatches = 0;
#ifdef ZSTD_PROF_TOP
    unsigned long long ztop[4] = {0, 0, 0, 0};
#endif
    unsigned long long zcnt[6] = {0, 0, 0, 0, 0, 0};     // chain outcomes: restart, continue, event; one-match batches; immediate repcodes; refreshes
#define ZCOUNT(k) (zcnt[k]++)
#else
#define ZCOUNT(k) ((void)0)
#endif
    if (n >= 20) {
        const u32 ilimit = n - 16;                                      // iend - HASH_READ_SIZE - PREFETCH_SAFETY (:448)
#ifdef AOCLGPU_EXPERIMENTS
        const bool chain = !(expf & 32) && sched == nullptr;            // experiment build: one match per batch everywhere
#else
        const bool chain = sched == nullptr; (void)expf;                // (chain batches are 64 consecutive probes: step 2)
#endif
        // Source ring of the chain batches.  Block b of the chunk (240 bytes) sits at ring offset (b & 1) * 240, the first
        // 32 bytes of the even block are mirrored behind the odd one so that a 32-byte read never wraps; the block after
        // the newest one is always in flight in a register (requested a block of parsing ahead of its use), so the 64
        // probe windows of a batch are LDS reads instead of a global round trip.
        lbyte* ring = lds + sizeof(T) * tsize + (kHi ? tsize / 8 : 0);
        int32_t rb = -4;                                                // newest block in the ring (rb-1 and rb are valid)
        u32 pend = 0; bool have_pend = false;                           // block rb+1, in flight
        auto blk_load = [&](int32_t b) -> u32 {
            const u32 a = bs + kRingBlk * (u32)b + 4u * (u32)lane;     // (blocks of the ring are counted from the block's start)
            u32 v = 0;
            if (lane < 60 && a + 4 <= n) v = ld32(src + a);             // (the chain never looks at the chunk's last 25 bytes)
            return v;
        };
        auto blk_store = [&](int32_t b, u32 v) {
            if (lane < 60) *(L32*)(ring + ((u32)b & 1u) * kRingBlk + 4u * (u32)lane) = v;
            if (!(b & 1) && lane < 8) *(L32*)(ring + 2 * kRingBlk + 4u * (u32)lane) = v;
        };
        // 8 bytes at chunk position P: from the ring when both ends lie in its two blocks (P is wave-uniform)
        auto ring_or_mem64 = [&](u32 P) -> u64 {
#ifndef ZSTD_NO_RING_TAIL
            const u32 r0 = P - bs, r1 = r0 + 7;
            const int32_t blo = (int32_t)(((u64)r0 * 17895698ull) >> 32), bhi = (int32_t)(((u64)r1 * 17895698ull) >> 32);   // / 240
            if (rb >= 0 && blo >= rb - 1 && bhi <= rb) return lds_ld64(ring + ((u32)blo & 1u) * kRingBlk + (r0 - kRingBlk * (u32)blo));
#endif
            return ld64(src + P);
        };
        // immediate rep2 matches behind a sequence that ends at ip0 (:690-702); returns where the next search starts
        auto rep_loop = [&](u32 ip0) -> u32 {
            bool hit = uni(ld32(src + ip0) == ld32(src + ip0 - rep2) ? 1u : 0u) != 0;
            while (hit) {
                ZCOUNT(4);
                const u32 rLength = wave_count(src, ip0 + 4, ip0 + 4 - rep2, n, lane) + 4;
                const u32 t = rep2; rep2 = rep1; rep1 = t;
                const u64 vc = ld64(src + ip0);
                const u32 hc = z_hash<MLS>(vc, hlog);
                if (lane == 0) tput(hc, ip0 + 2);
                put_seq(0, 1, rLength);
                wave_fence();
                ip0 += rLength;
                hit = ip0 <= ilimit && uni(ld32(src + ip0) == ld32(src + ip0 - rep2) ? 1u : 0u) != 0;
            }
            return ip0;
        };
        // State of the newest chain batch.  A search that starts inside it - behind a sequence of the exact path, behind the
        // immediate rep2 matches of rep_loop - is RESUMED in the same batch: every table write since then was made for a lane
        // in front of that search, and the lanes below lim have slots of their own, so what they read at the start of the
        // batch is what the reference reads; only the equality maps of the current repcodes have to be loaded (one round).
#ifndef ZSTD_RESUME_MARGIN
#define ZSTD_RESUME_MARGIN 8
#endif
        bool cbv = false;
        u32 b_pos0 = 0, b_h = 0, b_ent = 0, b_info = 0, b_eqb = 0, b_off = 0, b_pb0 = 0, b_flags = 0;   // flags: 1 match below lim, 2 hard
        int b_lim = 0; u64 b_later = 0;                                 // b_later: later lanes of the batch on this lane's slot
        for (;;) {                                                      // _start
#ifdef ZSTD_PROF_TOP
#define ZTT(v) unsigned long long v; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
            ZTT(tt0);
#endif
            u32 i0 = 0; int ev = 0;                                     // 1 = table match, 2 = repcode
            u32 e_pos = 0, e_cand = 0, e_sig = 0, e_st = 0, e_half = 0, e_p1 = 0, e_h1 = 0, e_h2 = 0;
            bool e_ok1 = false, e_ok2 = false;
            bool resume = false;
            if (cbv) {
                const u32 d = uni(s - b_pos0);                          // (s > b_pos0: at least one sequence since the batch began)
                resume = d + ZSTD_RESUME_MARGIN <= (u32)b_lim && (ballot((b_flags & 1u) != 0) >> d) != 0;
            }
            cbv = false;
            if (resume || (chain && s >= bs + 4 && (!FRAME || rep1 > 0) && (u64)s + 100 < ilimit)) {
                // ---------------- chain batch: the 64 consecutive probes a search starts with; every sequence they
                //                  hold is resolved from registers by the scalar walk of zstd_chain.h ----------------
                asm volatile("; batch begin");
#ifdef ZSTD_PROF_TOP
                ZTT(tt1);
                ztop[0] += tt1 - tt0;
#endif
                u64 e1 = 0, e2 = 0;
                if (!resume) {
                    const u32 pos0 = uni(s), pos = pos0 + (u32)lane;
                    typedef u32x4 __attribute__((aligned(1))) U4;
                    typedef const __attribute__((address_space(1))) U4* GP4;
                    // probe windows [pos-4, pos+28) from the source ring
                    const u32 rel0 = pos0 - 4 - bs;
                    const int32_t bl = (int32_t)(((u64)rel0 * 17895698ull) >> 32), bh = (int32_t)(((u64)(rel0 + 94) * 17895698ull) >> 32);   // / 240
                    if (!(bh <= rb && rb <= bl + 1)) {
                        if (have_pend && rb + 1 == bh && bl >= rb) { blk_store(rb + 1, pend); rb++; }
                        else {
                            const u32 v0 = blk_load(bl), v1 = blk_load(bh);
                            blk_store(bl, v0);
                            if (bh != bl) blk_store(bh, v1);
                            rb = bh;
                        }
                        pend = blk_load(rb + 1); have_pend = true;
                        wave_fence();
                    }
#ifdef ZSTD_PROF_TOP
                    ZTT(tt2);
                    ztop[1] += tt2 - tt1;
#endif
                    u32 ro = rel0 - kRingBlk * (u32)bl + ((u32)bl & 1u) * kRingBlk + (u32)lane;
                    if (ro >= 2 * kRingBlk) ro -= 2 * kRingBlk;
                    const u32x4 pa = lds_ld128(ring + ro), pb = lds_ld128(ring + ro + 16);
                    // byte equality at the two repcode distances over the 64 lanes: one byte load per lane each
                    const u32 b1 = src[pos - rep1];
                    u32 b2 = 0x100;
                    if (rep2) b2 = src[pos - rep2];
#ifdef ZSTD_PROF_TOP
                    { ZTT(tt3);
                      ztop[2] += tt3 - tt2; ztop[3]++; }
#endif
                    ZSTAGE(0);
#ifdef ZSTD_PROF
                    zbatches++;
#endif
                asm volatile("; probe begin");
                    const u64 seq = pa.y | ((u64)pa.z << 32);
                    const u32 h = z_hash<MLS>(seq, hlog);
                    const u32 ent = tget(h);
                    const bool has = ent >= prefixIdx;
                    const u32 c = ent - 2;
                    const bool near0 = c - prefixPos < 4;               // candidate within 4 bytes of the chunk / prefix start: exact path
                    u32x4 ca = {0, 0, 0, 0}, cb = {0, 0, 0, 0};
                    if (has) { const u32 a0 = near0 ? c : c - 4; ca = *(GP4)(src + a0); cb = *(GP4)(src + a0 + 16); }   // candidate window [c-4, c+28)
                    wave_fence();
                    ZSTAGE(1);
                asm volatile("; slots begin");
                    const u32 w = lds_first_lane_per_slot((u32)(uintptr_t)&table[h], (u32)lane);
                    wave_fence();
                    table[h] = (T)ent;                                    // low halves only: the extra bits were not touched
                    wave_fence();
                    ZSTAGE(2);
#ifdef ZSTD_PREFETCH
                    {   // warm the cache with the NEXT batch's candidates (positions pos0+64 ..): their table entries are known now
                        // (bar the few slots this batch commits), and the line is then on its way while this batch is resolved
                        pfacc ^= pfv;
                        const u32 pn = pos + 64;
                        if (pn + 8 <= n) {
                            const u32 en = tget(z_hash<MLS>(ld64(src + pn), hlog));
                            if (en >= prefixIdx + 2) pfv = src[en - 2];
                        }
                    }
#endif
                asm volatile("; masks begin");
#if defined(AOCLGPU_EXPERIMENTS) && defined(ZSTD_EXP_NOCUT)   /* timing only: what the cut at the first in-batch slot reuse costs (wrong parse) */
                    const int lim = 64; (void)w; const u64 later = 0;
#elif defined(ZSTD_CUT_ANY_DUP)                               /* round 4: the batch ends at ANY slot reuse */
                    const int lim = first_lane(ballot(w != (u32)lane)); const u64 later = 0;
#else
                    // Round 5: a lane whose slot an earlier lane of the batch names has a candidate that depends on the parse (that
                    // lane's position if its table write happens, else the table's entry) - but it is still decided when NEITHER can
                    // match: its own table candidate fails the 4-byte test and no earlier lane on its slot holds the same 4 bytes.  It
                    // stays in the batch as a probe that finds nothing; the batch is cut at the first duplicate that could match (the
                    // rule LZ4 and Snappy got in round 2).  `later`: the later lanes that name this lane's slot - when the batch commits,
                    // a slot goes to the last of its lanes that writes.
                    int lim = 64; u64 later = 0;
                    {
                        u64 dm = ballot(w != (u32)lane);
                        if (dm) {
                            const u32 c4 = near0 ? ca.x : ca.y;                       // the 4 bytes at the candidate
                            const u64 okb = ballot(has && c4 == pa.y);
                            for (; dm; dm &= dm - 1) {
                                const int k = (int)__builtin_ctzll(dm);
                                const u64 same = ballot(h == lane_get(h, k)) & ((1ull << k) - 1);
                                if (((okb >> k) & 1) || (ballot(pa.y == lane_get(pa.y, k)) & same) != 0) { lim = k; break; }
                                if ((same >> lane) & 1) later |= 1ull << k;
                            }
                        }
                    }
#endif
                    using zchain::zero_bytes8;
                    const u32 eqb = zero_bytes8(pa.x ^ ca.x, pa.y ^ ca.y) | (zero_bytes8(pa.z ^ ca.z, pa.w ^ ca.w) << 8) |
                                    (zero_bytes8(pb.x ^ cb.x, pb.y ^ cb.y) << 16) | (zero_bytes8(pb.z ^ cb.z, pb.w ^ cb.w) << 24);
                    const u32 fw = (u32)__builtin_ctz(~(eqb >> 8)), bk4 = (u32)__builtin_clz(~(eqb << 28));
                    const bool ok = has && (near0 ? ca.x == pa.y : ((eqb >> 4) & 15u) == 15u);
                    const u32 offv = pos - c;
                    // lanes that take the exact path whatever the parse looks like in front of them: candidate within 4 bytes of the
                    // prefix start, a run that leaves the 28-byte window, table writes that would fall behind lane 63
#if defined(AOCLGPU_EXPERIMENTS) && defined(ZSTD_EXP_NOLONG)   /* timing only: what the exact path costs for matches that outrun the window (wrong parse: cut at 28 bytes) */
                    const bool hardl = near0 || ((u32)lane + 2 + fw >= (u32)lim && lane >= 62);
#else
                    const bool hardl = near0 || fw >= 24 || ((u32)lane + 2 + fw >= (u32)lim && lane >= 62);
#endif
                    b_pos0 = pos0; b_lim = lim; b_h = h; b_ent = ent; b_eqb = eqb; b_off = offv; b_pb0 = pa.y & 0xFFu;   // (the byte at pos)
                    b_later = later;
                    b_info = near0 ? zchain::pack_info(FRAME ? 0u : offv, 0, 0)                          // (frames: the offset travels apart)
                                   : zchain::pack_info(FRAME ? 0u : offv, fw, zchain::back_code(bk4, c - prefixPos));
                    b_flags = ((lane < lim && ok) ? 1u : 0u) | (hardl ? 2u : 0u);
                    e1 = ballot(b_pb0 == b1); e2 = ballot(b_pb0 == b2);
                    ZSTAGE(6);
                }
                asm volatile("; walk begin");
                const u32 pos0 = b_pos0, pos = pos0 + (u32)lane;
                const int lim = b_lim;
                const u64 okmask = ballot((b_flags & 1u) != 0), hard = ballot((b_flags & 2u) != 0);
                zchain::WalkOut wo;
                {
                    u64 c0 = 0;
                    int32_t ar = (int32_t)anchor - (int32_t)pos0; int S0 = resume ? (int)(s - pos0) : 0; u32 r1 = rep1, r2 = rep2;
                    bool need_maps = resume;                            // (a resumed search begins at the anchor: ar == S0)
                    for (;;) {
                        if (need_maps) {
                            // one load round for full-width maps of both repcodes (lanes below a repcode's reach are not needed)
                            u32 nb1 = 0x100, nb2 = 0x100;
                            if (pos >= r1) nb1 = src[pos - r1];
                            if (r2 && pos >= r2) nb2 = src[pos - r2];
                            e1 = ballot(b_pb0 == nb1); e2 = ballot(b_pb0 == nb2);
                        }
                        if (nbSeq - qbase > 47u) flush_seq(nbSeq - qbase);  // (a walk parks at most 16 records)
#ifdef ZSTD_WALK_CXX
                        if constexpr (FRAME)
                            wo = zchain::walk(okmask, hard, lim, pos0, ar, r1, r2, e1, e2,
                                [&](u32 j) -> u64 { return ((u64)lane_get(b_off, (int)j) << 32) | lane_get(b_info, (int)j); }, [&](u32 j) { return lane_get(b_eqb, (int)j); },
                                [&](u32 ll, u32 off, u32 ml) { park_seq(ll, off, ml); }, S0, c0);
                        else
                            wo = zchain::walk(okmask, hard, lim, pos0, ar, r1, r2, e1, e2,
                                [&](u32 j) { return lane_get(b_info, (int)j); }, [&](u32 j) { return lane_get(b_eqb, (int)j); },
                                [&](u32 ll, u32 off, u32 ml) { park_seq(ll, off, ml); }, S0, c0);
#else
                        wo = walk_scalar<FRAME>(okmask, hard, lim, ar, r1, r2, e1, e2, b_info, b_eqb, b_off, q_ll, q_off, q_ml, nbSeq, qbase, S0, c0);
#endif
                        if (wo.outcome == zchain::ZC_REFRESH) {
                            // the map of the current rep1 came from a match's window and ends too early: fresh maps instead of a
                            // whole new batch
                            ZCOUNT(5);
                            S0 = wo.S; c0 = wo.commit; ar = wo.arel; r1 = wo.rep1; r2 = wo.rep2; need_maps = true;
                            continue;
                        }
                        break;
                    }
                }
                asm volatile("; commit begin");
                ZSTAGE(7);
                if (((wo.commit >> lane) & 1) && (b_later & wo.commit) == 0) tput(b_h, pos + 2);   // (a slot goes to the last of its lanes that writes)
                wave_fence();
#ifdef ZSTD_TRACE
                if (lane == 0 && g_ztrace_n < 4096) {
                    unsigned long long* t = g_ztrace + 8 * g_ztrace_n++;
                    t[0] = pos0; t[1] = (u32)lim; t[2] = okmask; t[3] = wo.commit; t[4] = (u32)wo.outcome | ((u64)(u32)wo.S << 8) | ((u64)(u32)wo.ev << 16) | ((u64)(u32)wo.el << 24) | ((u64)(u32)wo.ntail << 32) | ((u64)wo.post << 40);
                    t[5] = e1; t[6] = e2; t[7] = ((u64)rep1 << 32) | rep2;
                }
#endif
                anchor = (u32)((int32_t)pos0 + wo.arel); rep1 = wo.rep1; rep2 = wo.rep2; s = pos0 + (u32)wo.S;
                ZSTAGE(3);
                ZCOUNT(wo.outcome == zchain::ZC_RESTART ? 0 : (wo.outcome == zchain::ZC_CONTINUE ? 1 : 2));
                asm volatile("; outcome begin");
                if (wo.outcome == zchain::ZC_RESTART) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) {                       // the chain's last table writes, one by one
                        if (t >= 3 - wo.ntail) {
                            const int idx = wo.tail[t];
                            // (behind lane 63 - a match that ends behind the batch: the bytes are in the source ring more often than
                            //  not; a trip to memory here is exposed, the next batch's loads queue up behind it)
                            const u32 hh = idx < 64 ? lane_get(b_h, idx) : z_hash<MLS>(ring_or_mem64(pos0 + (u32)idx), hlog);
                            if (lane == 0) tput(hh, pos0 + (u32)idx + 2);
                            wave_fence();
                        }
                    }
                    if (wo.post) { s = rep_loop(s); anchor = s; cbv = true; }
                    ZSTAGE(4);
                    continue;
                }
                if (wo.outcome == zchain::ZC_CONTINUE) i0 = (u32)(wo.R - wo.S);
                else {
                    const int el = wo.el;
                    ev = wo.ev; cbv = true;
                    e_pos = pos0 + (u32)el; e_cand = lane_get(b_ent, el); e_sig = 2; e_st = 2; e_half = (u32)(el - wo.S) & 1u;
                    e_ok1 = el + 1 < 64; e_p1 = e_pos + 1; e_h1 = lane_get(b_h, (el + 1) & 63);
                    e_ok2 = el + 2 < 64; e_h2 = lane_get(b_h, (el + 2) & 63);
                }
            }
            asm volatile("; onematch begin");
            if (!ev) for (;;) {
                ZCOUNT(3);
                const u32 K = i0 + (u32)lane, m = K >> 1, half = K & 1;
                u32 am, sg, st;
                fast_sched_at(sched, m, am, sg, st);
                const bool valid = (u64)s + am + sg + 1 < ilimit;       // ip3 < ilimit (:503, :631)
                const u32 pos = s + am + half;
                ZSTAGE(5);
                u64 seq = 0;
                if (valid) seq = ld64(src + pos);
                ZSTAGE(0);
#ifdef ZSTD_PROF
                zbatches++;
#endif
                const u32 h = z_hash<MLS>(seq, hlog);
                u32 ent = 0;
                if (valid) ent = tget(h);
                const u32 cand = FRAME ? ent : (ent & kPosMask);
                const u32 mychk = chk((u32)seq);
                bool rep_hit = false;
                if (valid && half == 0 && (!FRAME || rep1 > 0)) {       // repcode at ip2 (:519-542)
                    const u32 r = pos + sg;
                    rep_hit = ld32(src + r) == ld32(src + r - rep1);
                }
                u32 m4 = ~(u32)seq;
                if (valid && cand >= prefixIdx && (FRAME || ((ent ^ mychk) & ~kPosMask) == 0)) m4 = ld32(src + cand - 2);
                wave_fence();
                ZSTAGE(1);
                const u32 w = lds_first_lane_per_slot((u32)(uintptr_t)&table[h], (u32)lane, valid);
                wave_fence();
                ZSTAGE(2);
                const int f = first_lane(ballot(valid && w != (u32)lane));
                const int e = first_lane(ballot(!valid));
                const int lim = f < e ? f : e;
                const bool mat_hit = m4 == (u32)seq;
                const int el = first_lane(ballot(lane < lim && (rep_hit || mat_hit)));
                if (valid) table[h] = (T)ent;                           // low halves only: the extra bits were not touched
                wave_fence();
                const int ncommit = (el < 64) ? el + 1 : lim;
                if (lane < ncommit) tput(h, pos + 2);
                wave_fence();
                if (el < 64) {
                    ev = lane_get((u32)rep_hit, el) ? 2 : 1;
                    e_pos = lane_get(pos, el); e_cand = lane_get(cand, el);
                    e_sig = lane_get(sg, el); e_st = lane_get(st, el); e_half = lane_get(half, el);
                    // the hashes of ip1 and current0+2 are already in the neighbouring lanes of the batch
                    e_ok1 = el + 1 < 64 && lane_get((u32)valid, (el + 1) & 63) != 0;
                    e_p1 = lane_get(pos, (el + 1) & 63); e_h1 = lane_get(h, (el + 1) & 63);
                    e_ok2 = el + 2 < 64 && lane_get((u32)valid, (el + 2) & 63) != 0 && lane_get(pos, (el + 2) & 63) == e_pos + 2;
                    e_h2 = lane_get(h, (el + 2) & 63);
                    break;
                }
                if (e < 64 && e <= f) break;
                i0 += (u32)lim;
            }
            if (!ev) break;                                             // _cleanup
            asm volatile("; exact begin");
            ZSTAGE(3);
            // table write of ip1 that accompanies the hit (:537, :557, :586-598)
            {
                const u32 ip1 = e_half == 0 ? e_pos + 1 : e_pos - 1 + e_sig;
                if (e_half == 0 || e_st <= 4) {
                    const u32 h1 = (e_ok1 && e_p1 == ip1) ? e_h1 : z_hash<MLS>(ld64(src + ip1), hlog);
                    if (lane == 0) tput(h1, ip1 + 2);
                    wave_fence();
                }
            }
            const u32 current0 = e_pos;
            u32 ip0, match0, offcode, mLength;
            // Catch-up and forward extension: the first round of both compares (64 bytes back, 256 bytes forward) is
            // issued together, one global round trip instead of two; longer runs continue with the loops.
            u32 maxback;
            if (ev == 2) { ip0 = e_pos + e_sig; match0 = ip0 - rep1; offcode = 1; maxback = 1; }       // :519-542
            else {                                                      // _offset (:651-665)
                ip0 = e_pos; match0 = e_cand - 2;
                rep2 = rep1; rep1 = ip0 - match0; offcode = rep1 + 3;
                const u32 a = ip0 - anchor, b = match0 - prefixPos; maxback = a < b ? a : b;
            }
            u32 fwd;
            {
                const u32 fa = ip0 + 4, fb = match0 + 4;
                bool eqb = false;
                if ((u32)lane < maxback) eqb = src[ip0 - 1 - (u32)lane] == src[match0 - 1 - (u32)lane];
                const u32 o = 4u * (u32)lane, pa = fa + o;
                u32 x = 0xFFFFFFFFu;
                if (pa + 4 <= n) x = ld32(src + pa) ^ ld32(src + fb + o);
                else if (pa < n) {
                    x = 0;
                    for (u32 t = 0; t < n - pa; ++t) x |= (u32)(src[pa + t] ^ src[fb + o + t]) << (8 * t);
                    x |= 0xFFFFFFFFu << (8 * (n - pa));
                }
                u32 back = (u32)first_lane(ballot(!eqb));               // <= maxback for maxback <= 64
                if (back == 64) {
                    while (back < maxback) {
                        const u32 k = back + (u32)lane;
                        bool eq = false;
                        if (k < maxback) eq = src[ip0 - 1 - k] == src[match0 - 1 - k];
                        const int c = first_lane(ballot(!eq));
                        back += (u32)c;
                        if (c < 64) break;
                    }
                }
                const int l = first_lane(ballot(x != 0));
                if (l == 64) fwd = 256 + wave_count(src, fa + 256, fb + 256, n, lane);
                else fwd = 4u * (u32)l + ((u32)__builtin_ctz(lane_get(x, l)) >> 3);
                ip0 -= back; match0 -= back; mLength = 4 + back + fwd;   // _match (:669)
            }
            ZSTAGE(4);
            {
                const u32 ll = ip0 - anchor;
                put_seq(ll, offcode, mLength);
            }
            ip0 += mLength; anchor = ip0;
            if (ip0 <= ilimit) {                                        // :678-702
                // the two table writes and the immediate-repcode test read four places: one round of loads
                const bool chk = rep2 > 0;
                u64 va = 0;
                if (!e_ok2) va = ld64(src + current0 + 2);
                const u64 vb = ld64(src + ip0 - 2);
                u32 r0 = 0, r1 = 1;
                if (chk) { r0 = ld32(src + ip0); r1 = ld32(src + ip0 - rep2); }
                const u32 ha = e_ok2 ? e_h2 : z_hash<MLS>(va, hlog), hb2 = z_hash<MLS>(vb, hlog);
                if (lane == 0) { tput(ha, current0 + 2 + 2); tput(hb2, ip0 - 2 + 2); }
                wave_fence();
                if (chk && uni(r0 == r1 ? 1u : 0u) != 0) { ip0 = rep_loop(ip0); anchor = ip0; }
            }
            s = ip0;
            asm volatile("; exact end");
        }
    }
#ifdef ZSTD_PROF
#ifdef ZSTD_PROF_TOP
    if (lane == 0) for (int k = 0; k < 4; ++k) atomicAdd(&g_zstd_gprof[k], ztop[k]);
#endif
    if (lane == 0) { for (int k = 0; k < 8; ++k) atomicAdd(&g_zstd_prof[k], zacc[k]); atomicAdd(&g_zstd_prof[8], zbatches); atomicAdd(&g_zstd_prof[9], (unsigned long long)nbSeq);
                     for (int k = 0; k < 6; ++k) atomicAdd(&g_zstd_prof[10 + k], zcnt[k]); }
#endif
    if constexpr (FRAME) {
        saved2 = (saved1 != 0 && rep1 != 0) ? saved1 : saved2;           // :614-630
        *rep1io = rep1 ? rep1 : saved1;
        *rep2io = rep2 ? rep2 : saved2;
    }
    flush_seq(nbSeq - qbase);
#ifdef ZSTD_PREFETCH
    if ((pfacc ^ pfv) == 0x5A17C0DEu) ws->pad[0] = 1;                    // (keeps the prefetch loads alive)
#endif
    wave_fence();
    u32 seqLits = 0;
#ifdef ZSTD_PROF
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long tg0 = __builtin_amdgcn_s_memtime();
#endif
    const u32 lastLits = gather_literals(src, n, ws, nbSeq, anchor, lane, &seqLits);
#ifdef ZSTD_PROF
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (lane == 0) { atomicAdd(&g_zstd_eprof[0], __builtin_amdgcn_s_memtime() - tg0); atomicAdd(&g_zstd_eprof[12], 1ull); atomicAdd(&g_zstd_eprof[13], (unsigned long long)nbSeq); atomicAdd(&g_zstd_eprof[14], (unsigned long long)(seqLits + lastLits)); }
#endif
    *litSizeOut = seqLits + lastLits;
    return nbSeq;
}

// ------------------------------------------------------------------------------------------------
// Double-fast match finder (levels 3-4): AOCL_ZSTD_compressBlock_doubleFast_noDict_generic, zstd_double_fast.c:332-573
// (ilimit = iend - 8 - 19, :351).  Two tables: L hashes 8 bytes, S hashes minMatch bytes; 2^16-2^17 + 2^13-2^17 u32 entries -
// far beyond LDS - so both live in this workgroup's slice of the HBM workspace (relaxed agent-scope accesses: served by L2,
// never by a stale L1 line; a wave's vector memory operations reach L2 in order).
//
// Same exact speculation as the other parsers.  Between two sequences the reference probes a data-independent schedule of
// positions (1-byte steps for 256 bytes, then the step grows by one every 256 bytes, :403-473); lane k evaluates probe k:
// both hashes, both table entries, both candidates, the repcode test at position + 1.  The batch is cut at the first lane
// whose long or short slot an earlier lane of the batch names (LDS scratch rounds on 10 hash bits, then the full hashes);
// the first event in probe order wins (per probe: repcode, long match, short match - :425, :438, :456), and exactly the
// probes up to it write both tables.  What follows an event is wave-uniform: the "long match one position further" test
// reads the table after the commit, so it sees what the reference sees.
// ------------------------------------------------------------------------------------------------
constexpr u32 kDfSchedLen = 1800;               // probes of one search within a 128 KiB block (offset 131072 is passed at probe 1758)
struct DfSched {
    u32 a[kDfSchedLen + 1];
    constexpr DfSched() : a() {
        u32 ip = 0, step = 1, nextStep = 256, ip1 = 1;
        for (u32 m = 0; m <= kDfSchedLen; ++m) {
            a[m] = ip;
            if (ip1 >= nextStep) { step++; nextStep += 256; }
            ip = ip1; ip1 += step;
        }
    }
};
__device__ const DfSched g_dfs{};
constexpr u32 kDfScratch = 1024;                // u16 slots per hash for the slot-reuse rounds
#ifndef DF_FIRST_WIDTH
#define DF_FIRST_WIDTH 16                       // lanes probed by the first batch of a search
#endif

__device__ __forceinline__ u32 z_hash8(u64 v, u32 hlog) { return (u32)((v * 0xCF1BBCDCB7A56463ull) >> (64 - hlog)); }   // ZSTD_hash8


// FRAME = false: the only block of a fresh frame (tables cleared here, repeat offsets {1,4,8}, no window limit).
// FRAME = true: block [bs, be) of a frame - tables, window and repeat offsets come from the frame loop (zstd_frame_wave) and
// the offsets go back to it (:393-399, :478-486).
template <int MLS, bool FRAME = false, bool GATHER = true>
__device__ u32 dfast_block(const gbyte* __restrict__ src, const u32 be, const u32 hlogL, const u32 hlogS, u32* tabs, lbyte* lds,
                           Workspace* ws_, u32* litSizeOut, int lane, const u32 bs = 0, const FrameWindow* win = nullptr,
                           u32* rep1io = nullptr, u32* rep2io = nullptr)
{
    GWorkspace* ws = (GWorkspace*)ws_;
    typedef __attribute__((address_space(1))) u32 gu32;
    typedef __attribute__((address_space(3))) uint16_t lu16;
    const u32 n = be;                                        // iend
    gu32* L = (gu32*)tabs;
    gu32* S = L + (1u << hlogL);
    if constexpr (!FRAME) {   // fresh tables (ZSTD_reset_matchState clears them for a new frame)
        typedef u32 u4 __attribute__((ext_vector_type(4)));
        const u32 quads = ((1u << hlogL) + (1u << hlogS)) / 4;
        for (u32 i = (u32)lane; i < quads; i += kWave) ((__attribute__((address_space(1))) u4*)L)[i] = u4{0, 0, 0, 0};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    }
    u32 prefixIdx = 2;                                       // candidates need an index above it (:438, :456)
    if constexpr (FRAME) prefixIdx = lowest_prefix(*win, be + 2);
    const u32 prefixPos = prefixIdx - 2;
    auto tget = [&](gu32* t, u32 h) -> u32 { return __hip_atomic_load((u32*)(t + h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto tput = [&](gu32* t, u32 h, u32 v) { __hip_atomic_store((u32*)(t + h), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // Chunks: an entry is index (18 bits) | 14 check bits of the bytes at that position (the 8 bytes the long table compares, the
    // 4 bytes the short one does).  The check bits never change a decision - a candidate whose check bits differ cannot pass the
    // compare - they spare the probing lanes the scattered candidate load, which is what this kernel is made of.  Frames need
    // all 32 bits for the index.
    constexpr bool kChk = !FRAME;
    constexpr u32 kIdxMask = kChk ? 0x3FFFFu : 0xFFFFFFFFu;
    auto chkL = [&](u64 v) -> u32 { return kChk ? ((u32)((v * 0x9E3779B97F4A7C15ull) >> 50) << 18) : 0u; };
    auto chkS = [&](u64 v) -> u32 { return kChk ? (((u32)v * 2246822519u) & ~kIdxMask) : 0u; };
    lu16* scrL = (lu16*)lds; lu16* scrS = scrL + kDfScratch;
    u32 q_ll = 0, q_off = 0, q_ml = 0, q_src = 0;           // sequence records parked one per lane, written 64 at a time
    auto flush_seq = [&](u32 first, u32 count) {
        if ((u32)lane < count) { ws->seqLL[first + lane] = q_ll; ws->seqOff[first + lane] = q_off; ws->seqML[first + lane] = q_ml; ws->seqSrc[first + lane] = q_src; }
    };
    auto put_seq = [&](u32 idx, u32 ll, u32 off, u32 ml, u32 srcpos) {
        const bool me = (u32)lane == (idx & 63u);
        q_ll = me ? ll : q_ll; q_off = me ? off : q_off; q_ml = me ? ml : q_ml; q_src = me ? srcpos : q_src;
        if ((idx & 63u) == 63u) flush_seq(idx - 63u, 64);
    };
    // backward extension of a match at (ip, m): bytes equal before both, at most maxback (catch-up loops :446, :502, :510)
    auto back_count = [&](u32 ip, u32 m, u32 maxback) -> u32 {
        u32 back = 0;
        while (back < maxback) {
            const u32 k = back + (u32)lane;
            bool eq = false;
            if (k < maxback) eq = src[ip - 1 - k] == src[m - 1 - k];
            const int c = first_lane(ballot(!eq));
            back += (u32)c;
            if (c < 64) break;
        }
        return back < maxback ? back : maxback;
    };
    u32 nbSeq = 0, litSize = 0, anchor = bs, s = bs + (bs == prefixPos ? 1u : 0u), rep1 = 1, rep2 = 0;   // rep {1,4,8}: 4 > maxRep = 1 is parked (:393-399)
    u32 saved1 = 0, saved2 = 0;
    if constexpr (FRAME) {
        rep1 = *rep1io; rep2 = *rep2io;
        const u32 curr0 = s + 2, maxRep = curr0 - lowest_prefix(*win, curr0);
        if (rep2 > maxRep) { saved2 = rep2; rep2 = 0; }
        if (rep1 > maxRep) { saved1 = rep1; rep1 = 0; }
    }
    const int64_t ilimit = (int64_t)n - 8 - 19;
    u32 i0 = 0;                                              // probes of the current search already examined
#ifdef ZSTD_PROF
    unsigned long long dacc[6] = {0, 0, 0, 0, 0, 0}, dprev = __builtin_amdgcn_s_memtime(), dcnt[4] = {0, 0, 0, 0};
#define DFSTAGE(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); dacc[k] += t_ - dprev; dprev = t_; } while (0)
#define DFCOUNT(k) (dcnt[k]++)
#else
#define DFSTAGE(k) do { } while (0)
#define DFCOUNT(k) ((void)0)
#endif
    for (;;) {
        // ---------------- one batch of 64 speculative probes ----------------
        DFCOUNT(i0 == 0 ? 0 : 1);
        const u32 m = i0 + (u32)lane;
        u32 am, am1;
        if (m < 256) { am = m; am1 = m + 1; }                // (the schedule is the identity for the first 256 bytes)
        else if (m < kDfSchedLen) { am = g_dfs.a[m]; am1 = g_dfs.a[m + 1]; }
        else { am = 0x40000000u; am1 = 0x40000000u; }
        const bool valid = (int64_t)s + (int64_t)am1 <= ilimit;             // the loop reaches probe m (:405, :473)
        const u32 pos = s + am;
        // A search's first batch probes DF_FIRST_WIDTH lanes only: both tables live in HBM (0.75 MB per chunk), every probe is
        // two scattered table reads and up to three candidate reads, and the kernel runs at the memory system's rate for such
        // reads (~52 G/s, bench `fetch_calibration`) - while most searches end within a few probes (the literal run in front of
        // a match is short).  A search that gets further continues with full batches.  `act` = probed in this batch.
        const int width = i0 == 0 ? DF_FIRST_WIDTH : 64;
        const bool act = valid && lane < width;
        u64 v8 = 0;
        if (act) v8 = ld64(src + pos);
        DFSTAGE(0);
        const u32 hl = z_hash8(v8, hlogL), hs = z_hash<MLS>(v8, hlogS);
        u32 idxl = 0, idxs = 0; bool plausL = true, plausS = true;
        if (act) {
            const u32 eL = tget(L, hl), eS = tget(S, hs);
            idxl = eL & kIdxMask; idxs = eS & kIdxMask;
            plausL = (eL & ~kIdxMask) == chkL(v8); plausS = (eS & ~kIdxMask) == chkS(v8);
        }
        DFSTAGE(1);
        u64 cl = ~v8; u32 cs = ~(u32)v8, r4 = ~(u32)(v8 >> 8);
        if (act && idxl > prefixIdx && plausL) cl = ld64(src + idxl - 2);
        if (act && idxs > prefixIdx && plausS) cs = ld32(src + idxs - 2);
        if (act && rep1 > 0) r4 = ld32(src + pos + 1 - rep1);
        const bool repHit = r4 == (u32)(v8 >> 8), longHit = cl == v8, shortHit = cs == (u32)v8;
        wave_fence();
        DFSTAGE(2);
        const u32 wl = lds_first_lane_per_slot((u32)(uintptr_t)(scrL + (hl & (kDfScratch - 1))), (u32)lane, act);
        wave_fence();
        const u32 wsm = lds_first_lane_per_slot((u32)(uintptr_t)(scrS + (hs & (kDfScratch - 1))), (u32)lane, act);
        wave_fence();
        int f = 64;                                          // first lane whose long or short slot an earlier lane names
        for (u64 d = ballot(act && (wl != (u32)lane || wsm != (u32)lane)); d; d &= d - 1) {
            const int j = (int)__builtin_ctzll(d);
            const u64 before = (1ull << j) - 1;
            if (((ballot(act && hl == lane_get(hl, j)) | ballot(act && hs == lane_get(hs, j))) & before) != 0) { f = j; break; }
        }
        const int e = first_lane(ballot(!valid));
        const int lim0 = f < e ? f : e, lim = lim0 < width ? lim0 : width;
        const int el = first_lane(ballot(lane < lim && (repHit || longHit || shortHit)));
        const int ncommit = el < 64 ? el + 1 : lim;
        DFSTAGE(3);
        if (lane < ncommit) { tput(L, hl, (pos + 2) | chkL(v8)); tput(S, hs, (pos + 2) | chkS(v8)); }   // hashLong[hl0] = hashSmall[hs0] = curr (:423)
        wave_fence();
        if (el >= 64) {
            if (e < 64 && e <= f && e <= width) break;       // the next probe's successor would pass ilimit: _cleanup
            i0 += (u32)lim;
            continue;
        }
        // ---------------- the event, wave-uniform from here ----------------
        // Every dependent global round trip below is a microsecond of this wave's life, so what one trip can fetch it does:
        // the probe after the event lane already holds ip1's bytes, long-table entry and candidate; the first rounds of the
        // forward and the backward compare travel together; so do the post-match insertions' bytes and the rep2 test's.
        const u32 curr = lane_get(pos, el), ip1 = s + lane_get(am1, el), stepE = lane_get(am1, el) - lane_get(am, el);
        const int kind = lane_get((u32)repHit, el) ? 2 : (lane_get((u32)longHit, el) ? 1 : 3);
        const bool nextOk = el + 1 < lim;                    // lane el+1 probed ip1 and no earlier lane of the batch names its slots
        const int eln = (el + 1) & 63;
        u32 ip = curr, mLength, mpos = 0, base = 4, offBase = 1;
        u32 hl1 = 0, ck1 = 0; bool insert1 = false;
        if (kind == 2) ip = curr + 1;                        // repcode at ip+1 (:425-431)
        else {
            u64 v1;
            if (nextOk) { v1 = ((u64)lane_get((u32)(v8 >> 32), eln) << 32) | lane_get((u32)v8, eln); hl1 = lane_get(hl, eln); }
            else { v1 = ld64(src + ip1); hl1 = uni(z_hash8(v1, hlogL)); }
            ck1 = chkL(v1);
            bool isLong = kind == 1;
            if (kind == 1) mpos = lane_get(idxl, el) - 2;    // long match at ip (:436-449)
            else {                                           // short match: a long match at ip1 wins (:495-504)
                u32 idxl1; u64 c1;
                if (nextOk) { idxl1 = lane_get(idxl, eln); c1 = ((u64)lane_get((u32)(cl >> 32), eln) << 32) | lane_get((u32)cl, eln); }
                else {
                    idxl1 = uni(tget(L, hl1)) & kIdxMask;
                    c1 = ~v1;
                    if (idxl1 > prefixIdx) c1 = ld64(src + idxl1 - 2);
                }
                if (idxl1 > prefixIdx && uni(c1 == v1 ? 1u : 0u)) { ip = ip1; mpos = idxl1 - 2; isLong = true; }
                else mpos = lane_get(idxs, el) - 2;
            }
            base = isLong ? 8u : 4u;
            insert1 = stepE < 4;
        }
        {   // forward extension (ZSTD_count) and catch-up: first rounds together
            const u32 m0 = kind == 2 ? ip - rep1 : mpos;
            const u32 fa = ip + base, fb = m0 + base;
            u32 maxback = 0;
            if (kind != 2) { const u32 a = ip - anchor, b2 = mpos - prefixPos; maxback = a < b2 ? a : b2; }
            bool eqb = false;
            if ((u32)lane < maxback) eqb = src[ip - 1 - (u32)lane] == src[m0 - 1 - (u32)lane];
            const u32 o = 4u * (u32)lane, pa = fa + o;
            u32 x = 0xFFFFFFFFu;
            if (pa + 4 <= n) x = ld32(src + pa) ^ ld32(src + fb + o);
            else if (pa < n) {
                x = 0;
                for (u32 t = 0; t < n - pa; ++t) x |= (u32)(src[pa + t] ^ src[fb + o + t]) << (8 * t);
                x |= 0xFFFFFFFFu << (8 * (n - pa));
            }
            u32 back = (u32)first_lane(ballot(!eqb));        // <= maxback for maxback <= 64
            if (back > maxback) back = maxback;
            if (back == 64 && maxback > 64) back = 64 + back_count(ip - 64, m0 - 64, maxback - 64);
            const int l = first_lane(ballot(x != 0));
            u32 fwd;
            if (l == 64) fwd = 256 + wave_count(src, fa + 256, fb + 256, n, lane);
            else fwd = 4u * (u32)l + ((u32)__builtin_ctz(lane_get(x, l)) >> 3);
            mLength = base + fwd + back;
            if (kind != 2) {
                const u32 offset = ip - mpos;
                ip -= back;
                rep2 = rep1; rep1 = offset; offBase = offset + 3u;
            }
        }
        DFSTAGE(4);
        if (insert1) { if (lane == 0) tput(L, hl1, (ip1 + 2) | ck1); }       // :516-526
        put_seq(nbSeq, ip - anchor, offBase, mLength, anchor);
        litSize += ip - anchor; nbSeq++;
        ip += mLength; anchor = ip;                          // _match_stored
        if ((int64_t)ip <= ilimit) {
            const u32 ins = curr + 2;                        // complementary insertions (:550-554), in the reference's order
            const bool chk = rep2 > 0;
            const u64 va = ld64(src + ins), vb = ld64(src + ip - 2), vc = ld64(src + ip - 1);
            u32 r0 = 0, r1 = 1;
            if (chk) { r0 = ld32(src + ip); r1 = ld32(src + ip - rep2); }   // the immediate rep2 test's bytes ride along
            if (lane == 0) {
                tput(L, z_hash8(va, hlogL), (ins + 2) | chkL(va));
                tput(L, z_hash8(vb, hlogL), (ip - 2 + 2) | chkL(vb));
                tput(S, z_hash<MLS>(va, hlogS), (ins + 2) | chkS(va));
                tput(S, z_hash<MLS>(vc, hlogS), (ip - 1 + 2) | chkS(vc));
            }
            wave_fence();
            bool hit = chk && uni(r0 == r1 ? 1u : 0u) != 0;
            while (hit) {                                    // :558-572
                const u32 rLength = wave_count(src, ip + 4, ip + 4 - rep2, n, lane) + 4;
                const u32 t = rep2; rep2 = rep1; rep1 = t;
                const u64 vi = ld64(src + ip);
                if (lane == 0) { tput(S, z_hash<MLS>(vi, hlogS), (ip + 2) | chkS(vi)); tput(L, z_hash8(vi, hlogL), (ip + 2) | chkL(vi)); }
                put_seq(nbSeq, 0, 1u, rLength, ip);
                wave_fence();
                nbSeq++;
                ip += rLength; anchor = ip;
                hit = (int64_t)ip <= ilimit && rep2 > 0 && uni(ld32(src + ip) == ld32(src + ip - rep2) ? 1u : 0u) != 0;
            }
        }
        s = ip; i0 = 0;
        DFSTAGE(5);
    }
#ifdef ZSTD_PROF
    if (lane == 0) { for (int k = 0; k < 6; ++k) atomicAdd(&g_zstd_prof[k], dacc[k]); atomicAdd(&g_zstd_prof[8], dcnt[0] + dcnt[1]); atomicAdd(&g_zstd_prof[9], (unsigned long long)nbSeq);
                     atomicAdd(&g_zstd_prof[10], dcnt[0]); atomicAdd(&g_zstd_prof[11], dcnt[1]); }
#endif
    if constexpr (FRAME) {
        saved2 = (saved1 != 0 && rep1 != 0) ? saved1 : saved2;           // :478-486
        *rep1io = rep1 ? rep1 : saved1;
        *rep2io = rep2 ? rep2 : saved2;
    }
    flush_seq(nbSeq & ~63u, nbSeq & 63u);
    wave_fence();
    if constexpr (GATHER) *litSizeOut = litSize + gather_literals(src, n, ws, nbSeq, anchor, lane);
    else *litSizeOut = litSize + (n - anchor);                           // (the literal buffer is filled by the kernel that codes the block)
    return nbSeq;
}

// ------------------------------------------------------------------------------------------------
// literals
// ------------------------------------------------------------------------------------------------
__device__ u32 lit_raw(g8* dst, const g8* src, u32 n, u32 type, int lane)
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
__device__ void wave_hist(l32* hist, const g8* src, u32 n, int lane)
{   // 16 bytes per lane and step (one load instead of sixteen round trips of one byte each), counts by LDS atomics; the next
    // step's load travels while this step counts
    typedef u32x4 __attribute__((aligned(1))) U;
    for (u32 i = (u32)lane; i < 256; i += kWave) hist[i] = 0;
    wave_fence();
    const u32 nv = n >> 4;
    u32x4 v = {0u, 0u, 0u, 0u};
    if ((u32)lane < nv) v = *(const ZGLB U*)(src + 16u * (u32)lane);
    for (u32 i = (u32)lane; i < nv; i += kWave) {
        u32x4 nx = {0u, 0u, 0u, 0u};
        if (i + kWave < nv) nx = *(const ZGLB U*)(src + 16u * (i + kWave));
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lds_add(&hist[w[k] & 0xFF], 1u);
            lds_add(&hist[(w[k] >> 8) & 0xFF], 1u);
            lds_add(&hist[(w[k] >> 16) & 0xFF], 1u);
            lds_add(&hist[w[k] >> 24], 1u);
        }
        v = nx;
    }
    for (u32 i = 16u * nv + (u32)lane; i < n; i += kWave) lds_add(&hist[src[i]], 1u);
    wave_fence();
}

// The code of the literals as one LDS word per symbol: value | length << 16 (one read per symbol in the stream loops).
template <class CT>
__device__ __forceinline__ void huf_code_words(l32* cw, CT ct, int lane)
{
#pragma unroll
    for (int t = 0; t < 4; ++t) { const u32 s = 64u * t + (u32)lane; cw[s] = (u32)ct->val[s] | ((u32)ct->nbBits[s] << 16); }
    wave_fence();
}

// HUF_compress1X_usingCTable (huf_compress.c:1051-1116): symbols last -> first, 8 symbols per lane per round.  The 8 symbols of
// a lane are ONE 8-byte load (they are consecutive bytes, taken from the top down), requested a round ahead.
__device__ u32 huf_encode_1x(g8* dst, const g8* src, u32 n, const l32* cw, l32* tile, int lane)
{
    WavePacker pk; pk.init(tile, dst, lane);
    // lane handles encode-order items [done + 8*lane, +8): item j is symbol src[n-1-j]: the bytes src[n-8-j0, n-j0) from the top down
    auto fetch = [&](u32 done) -> u64 {
        const u32 j0 = done + 8u * (u32)lane;
        if (j0 + 8 <= n) return ld64(src + (n - 8 - j0));
        u64 v = 0;
        if (j0 < n) { const u32 k = n - j0; for (u32 t = 0; t < k; ++t) v |= (u64)src[t] << (8 * (8 - k + t)); }   // the stream's first bytes: its last items
        return v;
    };
    u64 cur = fetch(0);
    for (u32 done = 0; done < n; done += 512) {
        const u64 nxt = done + 512 < n ? fetch(done + 512) : 0ull;
        const u32 j0 = done + 8u * (u32)lane;
        const u32 cnt = j0 >= n ? 0u : (n - j0 < 8 ? n - j0 : 8u);
        // two halves of four symbols (<= 44 bits each), then one 128-bit value
        u64 a = 0, b = 0; u32 na = 0, nb2 = 0;
#pragma unroll
        for (u32 k = 0; k < 4; ++k) if (k < cnt) { const u32 e = cw[(u32)(cur >> (8 * (7 - k))) & 0xFFu]; a |= (u64)(e & 0xFFFFu) << na; na += e >> 16; }
#pragma unroll
        for (u32 k = 4; k < 8; ++k) if (k < cnt) { const u32 e = cw[(u32)(cur >> (8 * (7 - k))) & 0xFFu]; b |= (u64)(e & 0xFFFFu) << nb2; nb2 += e >> 16; }
        const u64 lo = a | (b << na), hi = na ? b >> (64 - na) : 0ull;
        pk.push(lo, hi, na + nb2, lane);
        cur = nxt;
    }
    return pk.close(lane);
}
__device__ u32 huf_encode_4x(g8* dst_, const g8* src_, u32 n_, const l32* cw_, l32* tile_, int lane)
{   // huf_compress.c:1163-1211
    g8* dst = uni_ptr(dst_); const g8* src = uni_ptr(src_); const u32 n = uni(n_); const l32* cw = uni_ptr(cw_); l32* tile = uni_ptr(tile_);   // (a call: arguments arrive in vector registers)
    const u32 seg = (n + 3) / 4;
    u32 op = 6;
    if (n < 12) return 0;
    for (int i = 0; i < 3; ++i) {
        const u32 c = huf_encode_1x(dst + op, src + (u32)i * seg, seg, cw, tile, lane);
        if (c == 0 || c > 65535) return 0;
        if (lane == 0) { dst[2 * i] = (u8)c; dst[2 * i + 1] = (u8)(c >> 8); }
        op += c;
    }
    const u32 c = huf_encode_1x(dst + op, src + 3 * seg, n - 3 * seg, cw, tile, lane);
    if (c == 0 || c > 65535) return 0;
    return op + c;
}

// Huffman state a frame carries from block to block (ZSTD_hufCTables_t, zstd_compress_internal.h:70-73): the table of the
// last block that was written compressed and whether it may be reused (HUF_repeat_check; HUF_repeat_valid only comes from
// dictionaries).  `fresh` is set when the block just coded wrote a new table into EntropyLds::huf: the frame loop copies it
// into `prev` if the block is confirmed (ZSTD_blockState_confirmRepcodesAndEntropyTables, zstd_compress.c:4385-4387).
struct FrameHuf { ZLDS HufCT* prev; u32 repeat; u32 fresh; u32 disabled; };   // disabled: ZSTD_literalsCompressionIsDisabled

// HUF_estimateCompressedSize (huf_compress.c:809-818)
__device__ u32 huf_estimate(const ZLDS HufCT* ct, const l32* hist, u32 maxSym, int lane)
{
    u32 bits = 0;
    for (u32 i = (u32)lane; i <= maxSym; i += kWave) bits += (u32)ct->nbBits[i] * hist[i];
    return uni(wsum(bits)) >> 3;
}

// HUF_compress_internal (huf_compress.c:1326-1434): 0 = not compressible, 1 = rle.  FR: with the previous block's table
// (validation :1386-1391, small inputs prefer it :1392-1397, otherwise it is kept when it costs no more than a new table
// plus its description :1419-1426); `repeat` stays set exactly when the old table coded the literals.
template <bool FR>
__device__ u32 huf_compress(g8* dst, const g8* src, u32 n, bool single, bool suspect, LEntropy* L, int lane, FrameHuf* fh, u32& repeat)
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
    ESTAGE(L, 1);
    u32 largest = 0, maxSym = 0;
    for (u32 i = (u32)lane; i < 256; i += kWave) { const u32 c = L->hist[i]; if (c > largest) largest = c; if (c) maxSym = i; }
    for (int d = 32; d; d >>= 1) {
        const u32 t = __shfl_xor(largest, d); largest = largest > t ? largest : t;
        const u32 s2 = __shfl_xor(maxSym, d); maxSym = maxSym > s2 ? maxSym : s2;
    }
    if (largest == n) { if (lane == 0) dst[0] = src[0]; return 1; }
    if (largest <= (n >> 7) + 4) return 0;
    auto encode = [&](const ZLDS HufCT* ct, u32 h) -> u32 {              // HUF_compressCTable_internal :1216-1240
        huf_code_words(L->hist, ct, lane);                               // (the histogram is not needed any more)
        const u32 c = single ? huf_encode_1x(dst + h, src, n, L->hist, L->tile, lane) : huf_encode_4x(dst + h, src, n, L->hist, L->tile, lane);
        ESTAGE(L, 4);
        if (!c) return 0;
        if (h + c >= n - 1) return 0;
        return h + c;
    };
    if constexpr (FR) {
        if (repeat) {                                                    // HUF_validateCTable :796-807
            bool bad = false;
            for (u32 i = (u32)lane; i <= maxSym; i += kWave) bad |= L->hist[i] != 0 && fh->prev->nbBits[i] == 0;
            if (ballot(bad)) repeat = 0;
        }
        if (repeat && n <= 1024) return encode(fh->prev, 0);             // HUF_flags_preferRepeat (zstd_compress_literals.c:164)
    }
    u32 huffLog = fse_optimal_log(11, n, maxSym, 1);                     // HUF_optimalTableLog :1268-1283
    huffLog = huf_build_wave(&L->huf, L->hist, maxSym, huffLog, &L->hs, lane);
    ESTAGE(L, 2);
    const u32 h = huf_write_table_wave(dst, &L->huf, maxSym, huffLog, &L->hs, lane);
    wave_fence();
    ESTAGE(L, 3);
    if (!h) return 0;
    if constexpr (FR) {
        if (repeat) {
            const u32 oldSize = huf_estimate(fh->prev, L->hist, maxSym, lane), newSize = huf_estimate(&L->huf, L->hist, maxSym, lane);
            if (oldSize <= h + newSize || h + 12 >= n) return encode(fh->prev, 0);
        }
    }
    if (h + 12 >= n) return 0;
    if constexpr (FR) { repeat = 0; fh->fresh = 1; }
    return encode(&L->huf, h);
}

// ZSTD_compressLiterals (zstd_compress_literals.c:129-235)
template <bool FR>
__device__ u32 compress_literals(g8* dst, const g8* src, u32 n, bool suspect, LEntropy* L, int lane, FrameHuf* fh)
{
    const u32 lh = 3 + (n >= 1024) + (n >= 16384);
    const bool single = n < 256;
    u32 repeat = 0;
    if constexpr (FR) {
        repeat = fh->repeat; fh->fresh = 0;
        if (fh->disabled) return lit_raw(dst, src, n, 0, lane);          // fast with a target length (zstd_compress_internal.h:609-622)
    }
    if (n < 64) return lit_raw(dst, src, n, 0, lane);                    // ZSTD_minLiteralsToCompress = 8 << 3
    const u32 c = huf_compress<FR>(dst + lh, src, n, single, suspect, L, lane, fh, repeat);
    const u32 hType = repeat ? 3u : 2u;                                  // set_repeat : set_compressed (:185-189)
    if (c == 0 || c >= n - ((n >> 6) + 2)) { if constexpr (FR) fh->fresh = 0; return lit_raw(dst, src, n, 0, lane); }
    if (c == 1) { if constexpr (FR) fh->fresh = 0; return lit_raw(dst, src, n, 1, lane); }
    if (lane == 0) {
        if (lh == 3) { const u32 v = hType + ((u32)(!single) << 2) + (n << 4) + (c << 14); dst[0] = (u8)v; dst[1] = (u8)(v >> 8); dst[2] = (u8)(v >> 16); }
        else if (lh == 4) { const u32 v = hType + (2 << 2) + (n << 4) + (c << 18); st32(dst, v); }
        else { const u32 v = hType + (3 << 2) + (n << 4) + (c << 22); st32(dst, v); dst[4] = (u8)(c >> 10); }
    }
    return lh + c;
}

// ------------------------------------------------------------------------------------------------
// sequences
// ------------------------------------------------------------------------------------------------
// ZSTD_selectEncodingType, strategy fast, first block (zstd_compress_sequences.c:156-204): 0 basic, 1 rle, 2 compressed
__device__ int select_type(u32 mostFrequent, u32 nbSeq, u32 defaultNormLog, bool defaultAllowed, u32 strategy)
{
    if (mostFrequent == nbSeq) return (defaultAllowed && nbSeq <= 2) ? 0 : 1;
    if (defaultAllowed) {
        const u32 dynMin = ((1u << defaultNormLog) * (10 - strategy)) >> 3;               // mult = 10 - strategy (:184-186)
        if (nbSeq < dynMin || mostFrequent < (nbSeq >> (defaultNormLog - 1))) return 0;
    }
    return 2;
}

// mode + table description + CTable of one symbol stream from its histogram (ZSTD_buildCTable :242-288); first / last = the
// stream's codes of sequence 0 / nbSeq - 1.  All lanes call it; returns type (or -1), advances *op, updates *lastCount.
__device__ int build_stream(g8* base, u32* op, ZLDS FseCT* ct, const l32* count, const u32 first, const u32 last, u32 nbSeq, u32 maxCode, u32 fseLog,
                            const s16* defNorm, u32 defLog, u32 defMax, u32* lastCount, LEntropy* L, int lane, u32 strategy)
{
    // one symbol per lane (the alphabets have at most 53 symbols)
    int type = 0; u32 adv = 0, lc = 0;
    u32 c = (u32)lane <= maxCode ? count[lane] : 0u;
    const u32 max = (u32)top_lane(ballot(c != 0)), most = uni(wmaxu(c));
    const bool defaultAllowed = (defMax == 28) ? (max <= 28) : true;                  // zstd_compress.c:2847-2848
    type = select_type(most, nbSeq, defLog, defaultAllowed, strategy);
    ZLDS FseScratch* fs = &L->hs.fs;
    if (type == 1) {
        if (lane == 0) { fse_build_rle(ct, max); base[*op] = (u8)first; }
        adv = 1;
    } else if (type == 0) {
        const int norm = (u32)lane <= defMax ? (int)defNorm[lane] : 0;
        fse_build_ctable_wave(ct, norm, defMax, defLog, fs, lane);
    } else {
        u32 nbSeq1 = nbSeq;
        const u32 tableLog = fse_optimal_log(fseLog, nbSeq, max, 2);
        if (lane_get(c, (int)last) > 1) { if ((u32)lane == last) c--; nbSeq1--; }     // zstd_compress_sequences.c:264-268
        int norm = 0;
        if (fse_normalize_wave(norm, c, nbSeq1, max, tableLog, nbSeq1 >= 2048, lane) < 0) type = -1;
        else {
            const u32 h = fse_write_ncount_wave(base + *op, norm, max, tableLog, fs, lane);
            if (!h) type = -1;
            else { fse_build_ctable_wave(ct, norm, max, tableLog, fs, lane); adv = h; lc = h; }
        }
    }
    wave_fence();
    *op += adv;
    if (lc) *lastCount = lc;
    return type;
}

// ZSTD_entropyCompressSeqStore_internal (zstd_compress.c:2910-3027): block content size, 0 = emit raw
#ifndef ZSTD_FSE_WARM
#define ZSTD_FSE_WARM 128      /* steps a chain segment walks in front of its first step (a multiple of 16) */
#endif
template <bool FR = false>
__device__ u32 entropy_compress(u8* dst_, Workspace* ws_, u32 litSize_, u32 nbSeq_, EntropyLds* L_, int lane, u32 strategy_, FrameHuf* fh_ = nullptr)
{
    // (a call: the arguments arrive in vector registers and count as divergent - made scalar again, or every loop over the
    //  sequences below is an exec-mask loop with 64-bit vector address arithmetic; and they arrive as generic pointers: given
    //  their address spaces here, everything below is global_* / ds_*, not flat_*)
    g8* dst = (g8*)uni_ptr(dst_); GWorkspace* ws = (GWorkspace*)uni_ptr(ws_); const u32 litSize = uni(litSize_), nbSeq = uni(nbSeq_), strategy = uni(strategy_);
    LEntropy* L = (LEntropy*)uni_ptr(L_); FrameHuf* fh = uni_ptr(fh_);
    u32 op = 0;
    const bool suspect = (nbSeq == 0) || (litSize / nbSeq >= 20);
    ESTAGE(L, -1);
    op += compress_literals<FR>(dst, ws->lits, litSize, suspect, L, lane, fh);
    ESTAGE(L, 4);
    if (lane == 0) {
        if (nbSeq < 128) dst[op] = (u8)nbSeq;
        else if (nbSeq < 0x7F00) { dst[op] = (u8)((nbSeq >> 8) + 0x80); dst[op + 1] = (u8)nbSeq; }
        else { dst[op] = 0xFF; dst[op + 1] = (u8)(nbSeq - 0x7F00); dst[op + 2] = (u8)((nbSeq - 0x7F00) >> 8); }
    }
    op += nbSeq < 128 ? 1 : (nbSeq < 0x7F00 ? 2 : 3);
    if (nbSeq == 0) return op;
    const g32* gLL = ws->seqLL; const g32* gML = ws->seqML; const g32* gOff = ws->seqOff;
    g32* gCodes = ws->codes;
    // ---- codes (ZSTD_seqToCodes :2715-2741) and the three histograms in one pass over the sequences, four rounds of loads in
    //      flight (round 6; until then: one pass for the codes, one byte-load pass per stream for its histogram) ----
    for (u32 i = (u32)lane; i < 3 * 64; i += kWave) (&L->cnt3[0][0])[i] = 0;
    if (lane < 36) L->llbits[lane] = kLLbits[lane];
    if (lane < 53) L->mlbits[lane] = kMLbits[lane];
    wave_fence();
    for (u32 i0 = (u32)lane; i0 < nbSeq; i0 += 4 * kWave) {
        u32 ll[4], ml[4], of[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { const u32 i = i0 + (u32)t * kWave; ll[t] = 0; ml[t] = 3; of[t] = 1; if (i < nbSeq) { ll[t] = gLL[i]; ml[t] = gML[i]; of[t] = gOff[i]; } }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const u32 i = i0 + (u32)t * kWave;
            if (i < nbSeq) {
                const u32 m = ml[t] - 3;
                const u32 cl = ll[t] > 0xFFFF ? 35u : ll_code(ll[t]), cm = m > 0xFFFF ? 52u : ml_code(m), co = (u32)hb(of[t]);
                gCodes[i] = cl | (cm << 8) | (co << 16);
                lds_add(&L->cnt3[0][cl], 1u); lds_add(&L->cnt3[1][cm], 1u); lds_add(&L->cnt3[2][co], 1u);
            }
        }
    }
    wave_fence();
    ESTAGE(L, 5);
    const u32 seqHead = op++;
    u32 lastCount = 0;
    const u32 cwFirst = uni(gCodes[0]), cwLast = uni(gCodes[nbSeq - 1]);
    const int tLL = build_stream(dst, &op, &L->ctLL, L->cnt3[0], cwFirst & 0xFF, cwLast & 0xFF, nbSeq, 35, 9, kLLnorm, 6, 35, &lastCount, L, lane, strategy);
    if (tLL < 0) return 0;
    const int tOF = build_stream(dst, &op, &L->ctOF, L->cnt3[2], (cwFirst >> 16) & 0xFF, (cwLast >> 16) & 0xFF, nbSeq, 31, 8, kOFnorm, 5, 28, &lastCount, L, lane, strategy);
    if (tOF < 0) return 0;
    const int tML = build_stream(dst, &op, &L->ctML, L->cnt3[1], (cwFirst >> 8) & 0xFF, (cwLast >> 8) & 0xFF, nbSeq, 52, 9, kMLnorm, 6, 52, &lastCount, L, lane, strategy);
    if (tML < 0) return 0;
    if (lane == 0) dst[seqHead] = (u8)((tLL << 6) + (tOF << 4) + (tML << 2));
    ESTAGE(L, 6);

    // ZSTD_encodeSequences_body (zstd_compress_sequences.c:290-382).  Encode step e = 0..nbSeq-1 handles sequence
    // i = nbSeq-1-e; step 0 only initialises the three states, every later step emits the low bits of a state and moves it:
    //     nb = (state + deltaNbBits[code]) >> 16;  bits = state & mask(nb);  state = stateTable[(state >> nb) + deltaFindState[code]].
    // ---- phase A: the state chains, in segments (round 6).  Until round 5 lanes 0-2 walked the three chains from end to end, one
    //      dependent LDS read per step: 28 % of the entropy stage.  The new state depends on state >> nb only - after a step the
    //      chain remembers log2(states of the symbol) bits of where it came from - so a chain started from ANY state of the
    //      right symbol falls into step with the true one within a few dozen steps.  Each stream gets 21 lanes; lane g walks steps
    //      [b_g, b_g + seg) after a warm-up of 64 steps that starts from the symbol's initial state (for g = 0, whose warm-up
    //      is empty, that IS the true state), and records the state BEFORE every step (two bytes in the workspace; what the
    //      step emits follows from it).  Then the proof: lane g is right when the state it entered its segment with equals the
    //      state lane g-1 left with; a lane that was wrong walks again from that state until it meets a state it recorded
    //      before (from there on nothing changes), and the check is repeated until every lane agrees with its predecessor -
    //      lane 0 is right by construction, so all are.  Worst case (no chain ever converges) = the serial walk.
    const u32 steps = nbSeq - 1;
    u32 fin = 0;                                                          // lane 0 / 21 / 42: the stream's state after the last step
    {
        constexpr u32 kWarm = ZSTD_FSE_WARM, kLanes = 21;
        const u32 grp = (u32)lane / kLanes, g = (u32)lane - kLanes * grp;   // grp 0 LL, 1 ML, 2 OF (3: lane 63, idle)
        const ZLDS FseCT* ct = grp == 0 ? &L->ctLL : (grp == 1 ? &L->ctML : &L->ctOF);
        const u32 sh = 8u * (grp < 3 ? grp : 0u);
        g16* gSt = ws->fseState[grp < 3 ? grp : 0u];
        u32 G = steps / 128; G = G < 1 ? 1u : (G > kLanes ? kLanes : G);       // (segments of at least 128 steps)
        const u32 seg = ((steps + G - 1) / G + 15u) & ~15u;               // a multiple of sixteen steps (one block)
        const u32 b = 1 + g * seg;                                        // first step of the lane's segment
        const bool on = grp < 3 && b <= steps;
        const u32 e_end = on ? (b + seg < nbSeq ? b + seg : nbSeq) : 0u;  // one past its last step
        const u32 w = b > kWarm ? b - kWarm : 1u;                         // first step of its warm-up (w == b for lane 0)
        typedef u32x4 __attribute__((aligned(1))) U128;
        // code words of steps e .. e+3 = sequences nbSeq-1-e .. nbSeq-4-e: one 16-byte load, .w first (the last block of a chain
        // may reach into codesPad: those steps are masked)
        auto codes4 = [&](u32 e, bool p) -> u32x4 {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (p) v = *(const ZGLB U128*)(gCodes + ((int)nbSeq - 4 - (int)e));
            return v;
        };
        // four steps.  Every LDS read is unconditional (a lane that is past its end reads entry 0 and keeps its state): no branch
        // in the chain; the four (deltaNbBits, deltaFindState) pairs are read in front of the four dependent state reads.
        auto step4 = [&](u32& st, const u32x4 cwv, const u32 e, const u32 lim, u32& r0, u32& r1) {
            const u32 cws[4] = {cwv.w, cwv.z, cwv.y, cwv.x};
            int dnb[4], dfs[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const u32 c = e + (u32)k < lim ? (cws[k] >> sh) & 0xFFu : 0u; dnb[k] = ct->deltaNbBits[c]; dfs[k] = ct->deltaFindState[c]; }
            u32 sv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool act = e + (u32)k < lim;
                sv[k] = st;
                const u32 nb = (st + (u32)dnb[k]) >> 16;
                const u32 nxs = ct->stateTable[act ? (int)(st >> nb) + dfs[k] : 0];
                st = act ? nxs : st;
            }
            r0 = sv[0] | (sv[1] << 16); r1 = sv[2] | (sv[3] << 16);
        };
        u32 st = 0, S = 0;
        if (grp < 3) {
            const u32 iw = on ? nbSeq - w : nbSeq - 1;                    // sequence of step w - 1 (lanes that are off: step 0's, for `fin`)
            st = fse_init_state(ct, (gCodes[iw] >> sh) & 0xFFu);
        }
        // Blocks of sixteen steps.  A block's four code loads are requested a block ahead, and its sixteen recorded states leave at
        // the top of the NEXT block - behind the wait for that block's codes: with loads and stores in flight together the wait
        // is for everything (they may return out of order), and this way everything it waits for is a block old.
        u32 e = on ? w : 0u;
        u32x4 c0 = codes4(e, e < e_end), c1 = codes4(e + 4, e + 4 < e_end), c2 = codes4(e + 8, e + 8 < e_end), c3 = codes4(e + 12, e + 12 < e_end);
        u32x4 pa = {0u, 0u, 0u, 0u}, pb = pa; u32 pe = 0; bool pst = false;  // the block whose states wait to be stored
        const u32 trips = (kWarm + seg) / 16;
        for (u32 t = 0; t <= trips; ++t) {
            asm volatile("" ::: "memory");
            if (pst) { *(ZGLB U128*)(gSt + pe) = pa; *(ZGLB U128*)(gSt + pe + 8) = pb; }
            const bool act = e < e_end;
            const u32 en = act ? e + 16 : e;
            const u32x4 n0 = codes4(en, act && en < e_end), n1 = codes4(en + 4, act && en + 4 < e_end), n2 = codes4(en + 8, act && en + 8 < e_end),
                        n3 = codes4(en + 12, act && en + 12 < e_end);
            if (act && e == b) S = st;
            const u32 lim = act ? e_end : 0u;
            u32 ra, rb, rc, rd;
            step4(st, c0, e, lim, ra, rb); step4(st, c1, e + 4, lim, rc, rd); pa = u32x4{ra, rb, rc, rd};
            step4(st, c2, e + 8, lim, ra, rb); step4(st, c3, e + 12, lim, rc, rd); pb = u32x4{ra, rb, rc, rd};
            pst = act && e >= b; pe = e;
            e = en; c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        }
        wave_fence();
        ESTAGE(L, 7);
        // the proof, and the repairs
        for (;;) {
            const u32 prev = wave_shr1(st);                               // the state lane g-1 left its segment with
            const bool bad = on && g > 0 && S != prev;
            if (!any_lane(bad)) break;
#ifdef ZSTD_PROF
            { const u32 nbad = (u32)__builtin_popcountll(ballot(bad)); if (lane == 0) atomicAdd(&g_zstd_eprof[15], (unsigned long long)nbad); }
#endif
            // the same blocks of sixteen steps: the next block's codes and its first recorded state travel while this block is
            // walked; a block's new states leave at the top of the next trip
            u32 e2 = bad ? b : e_end, st2 = prev;
            if (bad) S = prev;
            bool run = bad;
            auto rec16 = [&](u32 ee, bool p) -> u32 { u32 v = 0; if (p) v = (u32)gSt[ee]; return v; };
            u32x4 d0 = codes4(e2, run), d1 = codes4(e2 + 4, run && e2 + 4 < e_end), d2 = codes4(e2 + 8, run && e2 + 8 < e_end), d3 = codes4(e2 + 12, run && e2 + 12 < e_end);
            u32 rec = rec16(e2, run);
            u32x4 qa = {0u, 0u, 0u, 0u}, qb = qa; u32 qe = 0; bool qst = false;
            while (any_lane(run || qst)) {
                asm volatile("" ::: "memory");
                if (qst) { *(ZGLB U128*)(gSt + qe) = qa; *(ZGLB U128*)(gSt + qe + 8) = qb; }
                qst = false;
                if (run && rec == st2) run = false;                       // met the walk recorded before: the rest of the segment stands
                const u32 en = e2 + 16;
                const bool more = run && en < e_end;
                const u32x4 n0 = codes4(en, more), n1 = codes4(en + 4, more && en + 4 < e_end), n2 = codes4(en + 8, more && en + 8 < e_end),
                            n3 = codes4(en + 12, more && en + 12 < e_end);
                const u32 nrec = rec16(en, more);
                if (run) {
                    u32 ra, rb, rc, rd;
                    step4(st2, d0, e2, e_end, ra, rb); step4(st2, d1, e2 + 4, e_end, rc, rd); qa = u32x4{ra, rb, rc, rd};
                    step4(st2, d2, e2 + 8, e_end, ra, rb); step4(st2, d3, e2 + 12, e_end, rc, rd); qb = u32x4{ra, rb, rc, rd};
                    qst = true; qe = e2;
                    e2 = en;
                    if (e2 >= e_end) { st = st2; run = false; }           // walked to the end: a new final state
                }
                d0 = n0; d1 = n1; d2 = n2; d3 = n3; rec = nrec;
                wave_fence();
            }
        }
        // final states: the last lane of each stream that is on (or lane g = 0 when there are no steps at all)
        const u32 used = steps ? (steps + seg - 1) / seg : 1u;            // lanes per stream that walked
        fin = st;
        const u32 fLL = lane_get(st, (int)(used - 1)), fML = lane_get(st, (int)(kLanes + used - 1)), fOF = lane_get(st, (int)(2 * kLanes + used - 1));
        fin = lane == 0 ? fLL : (lane == 1 ? fML : fOF);
    }
    wave_fence();
    ESTAGE(L, 8);
    // ---- phase B: every lane packs one sequence per round {OF state bits, ML state bits, LL state bits, LL extra, ML extra,
    //      OF extra}; the next round's seven loads travel while this round is packed ----
    WavePacker pk; pk.init(L->tile, dst + op, lane);
    {
        const g16* sLLp = ws->fseState[0]; const g16* sMLp = ws->fseState[1]; const g16* sOFp = ws->fseState[2];
        struct Rec { u32 cw, ll, ml, of, sl, sm, so; };
        auto load = [&](u32 e0) -> Rec {
            Rec r = {0u, 0u, 3u, 0u, 0u, 0u, 0u};
            const u32 e = e0 + (u32)lane;
            if (e < nbSeq) {
                const u32 i = nbSeq - 1 - e;
                r.cw = gCodes[i]; r.ll = gLL[i]; r.ml = gML[i]; r.of = gOff[i];
                if (e > 0) { r.sl = sLLp[e]; r.sm = sMLp[e]; r.so = sOFp[e]; }
            }
            return r;
        };
        Rec cur = load(0);
        for (u32 e0 = 0; e0 < nbSeq; e0 += kWave) {
            const Rec nx = load(e0 + kWave);
            const u32 e = e0 + (u32)lane;
            u64 lo = 0, hi = 0; u32 nb = 0;
            if (e < nbSeq) {
                const u32 cl = cur.cw & 0xFF, cm = (cur.cw >> 8) & 0xFF, co = (cur.cw >> 16) & 0xFF;
                auto put = [&](u64 v, u32 l) {
                    if (!l) return;
                    v &= (1ull << l) - 1;
                    if (nb < 64) { lo |= v << nb; if (nb + l > 64) hi |= v >> (64 - nb); }
                    else hi |= v << (nb - 64);
                    nb += l;
                };
                if (e > 0) {
                    const u32 no = (cur.so + (u32)L->ctOF.deltaNbBits[co]) >> 16, nm = (cur.sm + (u32)L->ctML.deltaNbBits[cm]) >> 16,
                              nl = (cur.sl + (u32)L->ctLL.deltaNbBits[cl]) >> 16;
                    put(cur.so, no); put(cur.sm, nm); put(cur.sl, nl);
                }
                put(cur.ll, L->llbits[cl]); put(cur.ml - 3, L->mlbits[cm]); put(cur.of, co);
            }
            pk.push(lo, hi, nb, lane);
            cur = nx;
        }
    }
    ESTAGE(L, 9);
    {   // FSE_flushCState x3: ML, OF, LL (:371-373)
        const u32 sLL = lane_get(fin, 0), sML = lane_get(fin, 1), sOF = lane_get(fin, 2);
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
    ESTAGE(L, 10);
    if (lastCount && lastCount + bs < 4) return 0;                       // zstd_compress.c:3017-3023
    return op;
}

// cParams of levels 1..4 for a known size of at most 128 KiB: the rows of clevels.h:80-88,106-114 (upstream table) and
// :211-219,237-245 (the AOCL table the reference uses once its dispatcher has run: it differs in the double-fast rows) after
// ZSTD_adjustCParams_internal (zstd_compress.c:1575-1595: windowLog <= srcLog, hashLog <= windowLog + 1, chainLog <= windowLog).
// strategy 1 = fast (levels 1-2: one table of 2^hlog slots, step 2), 2 = double-fast (levels 3-4: long table 2^hlog, short
// table 2^clog), 0 = no GPU parse (level 4 is "greedy" for inputs of at most 16 KiB).
struct ZParams { u32 strategy, hlog, clog, mls; };
__device__ __forceinline__ ZParams level_cparams(u32 level, bool upstream, u32 n)
{
    const u32 srcLog = (n < 64) ? 6 : (u32)hb(n - 1) + 1;
    u32 W, C, H, M, strat;
    if (n <= 16384) {
        W = 14;
        if (level <= 1) { C = 14; H = 15; M = 5; strat = 1; }
        else if (level == 2) { C = 14; H = 15; M = 4; strat = 1; }
        else if (level == 3) { C = upstream ? 14 : 13; H = 15; M = 4; strat = 2; }
        else { C = 14; H = 14; M = 4; strat = 0; }
    } else {
        W = 17;
        if (level <= 1) { C = 12; H = 13; M = 6; strat = 1; }
        else if (level == 2) { C = 13; H = 15; M = 5; strat = 1; }
        else if (level == 3) { C = upstream ? 15 : 14; H = 16; M = 5; strat = 2; }
        else { C = upstream ? 17 : 15; H = 17; M = 4; strat = 2; }
    }
    const u32 w = srcLog < W ? srcLog : W;
    ZParams p;
    p.strategy = strat; p.hlog = H < w + 1 ? H : w + 1; p.clog = C < w ? C : w; p.mls = M;
    return p;
}
__device__ __forceinline__ void level_params(u32 level, u32 n, u32* hlog, u32* mls)
{
    const ZParams p = level_cparams(level, false, n);
    *hlog = p.hlog; *mls = p.mls;
}
__device__ __forceinline__ void level1_params(u32 n, u32* hlog, u32* mls) { level_params(1, n, hlog, mls); }
__device__ __forceinline__ u32 flags_level(u32 flags)
{   // AOCLGPU_ZSTD_LEVEL(l) in bits 16-18 (0 = not given); AOCLGPU_FLAG_ZSTD_LEVEL2 is the older spelling of level 2
    const u32 l = (flags >> 16) & 7u;
    return l ? l : ((flags & AOCLGPU_FLAG_ZSTD_LEVEL2) ? 2u : 1u);
}

// DF = the double-fast launches (levels 3-4).  Two kernels on purpose: the register budget of the level 1-2 kernel decides how
// many chunks are resident per CU (136 VGPRs -> 9 by LDS; with the double-fast finder compiled into the same kernel it needed
// 181 -> 8 per CU and level 1 lost 10 %).
template <bool DF, int PHASE = 0>
__device__ int zstd_compress_wave(const u8* __restrict__ src, const u32 n, u8* __restrict__ dst, const u32 cap,
                                  void* lds, u32 ldsBytes, Workspace* ws, u32* tabs, int lane, u32 expf, void* entLds = nullptr, u32* lock = nullptr,
                                  u32* meta = nullptr)
{   // entLds / lock (double-fast kernel): the entropy stage's LDS is shared by the workgroup's wavefronts, taken one at a time
    // PHASE (double-fast as two kernels): 1 = the match finder alone, its sequence count and literal size go to meta[0..1];
    // 2 = everything else, from the sequences and literals phase 1 left in the workspace
    // (the block is staged in the workspace and only copied out when it fits: a slot smaller than ZSTD_compressBound works
    //  whenever the frame itself fits, as with the reference's ZSTD_compress)
    if (n > kBlockMax || cap < 5u + ((n < 256) ? 1u : (n < 65536 + 256 ? 2u : 4u)) + 3u) return 0;
    const ZParams zp = level_cparams(flags_level(expf), (expf & AOCLGPU_FLAG_ZSTD_UPSTREAM_TABLE) != 0, n);
    if (zp.strategy == 0 || (zp.strategy == 2) != DF || (DF && PHASE != 2 && !tabs)) return 0;   // no GPU parse for these parameters / wrong launch class
    if constexpr (PHASE == 1) {
        static_assert(DF || PHASE != 1, "only the double-fast finder runs as a kernel of its own");
        if (n >= 7) {
            u32 litSize = 0, nbSeq = 0;
            if (2u * 2u * kDfScratch > ldsBytes) return 0;
            if (zp.mls == 4) nbSeq = dfast_block<4, false, false>((const gbyte*)src, n, zp.hlog, zp.clog, tabs, (lbyte*)lds, ws, &litSize, lane);
            else nbSeq = dfast_block<5, false, false>((const gbyte*)src, n, zp.hlog, zp.clog, tabs, (lbyte*)lds, ws, &litSize, lane);
            if (lane == 0) { meta[0] = nbSeq; meta[1] = litSize; }
        }
        return 0;
    }
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
        u32 litSize = 0, nbSeq;
        const u32 hlog = zp.hlog, mls = zp.mls;
        if constexpr (PHASE == 2) {
            nbSeq = uni(meta[0]); litSize = uni(meta[1]); (void)hlog; (void)mls;
            (void)gather_literals((const gbyte*)src, n, (GWorkspace*)ws, nbSeq, 0xFFFFFFFFu, lane, nullptr, litSize);
        }
        else if constexpr (DF) {
            if (2u * 2u * kDfScratch > ldsBytes) return 0;
            if (mls == 4) nbSeq = dfast_block<4>((const gbyte*)src, n, hlog, zp.clog, tabs, (lbyte*)lds, ws, &litSize, lane);
            else nbSeq = dfast_block<5>((const gbyte*)src, n, hlog, zp.clog, tabs, (lbyte*)lds, ws, &litSize, lane);
        } else {
            const bool hi = n > 16384;
            const u32 need = (2u << hlog) + (hi ? (1u << hlog) / 8 : 0u) + kSrcRingBytes;
            if (need > ldsBytes) return 0;                               // launcher sizes LDS per chunk class
            if (mls == 4) nbSeq = fast_block<4, u16, false>((const gbyte*)src, n, hlog, (lbyte*)lds, ws, &litSize, lane, expf);
            else if (mls == 6) nbSeq = fast_block<6, u16, true>((const gbyte*)src, n, hlog, (lbyte*)lds, ws, &litSize, lane, expf);
            else if (hi) nbSeq = fast_block<5, u16, true>((const gbyte*)src, n, hlog, (lbyte*)lds, ws, &litSize, lane, expf);
            else nbSeq = fast_block<5, u16, false>((const gbyte*)src, n, hlog, (lbyte*)lds, ws, &litSize, lane, expf);
        }
        wave_fence();
#ifdef AOCLGPU_EXPERIMENTS
        if (expf & 16) cSize = 0;      // experiment build (timing only): match finder alone, block emitted raw
        else
#endif
        {
            if (lock) {
                if (lane == 0) while (atomicCAS(lock, 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(32);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            cSize = entropy_compress(ws->tmp, ws, litSize, nbSeq, reinterpret_cast<EntropyLds*>(entLds ? entLds : lds), lane, zp.strategy);
            if (lock) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) atomicExch(lock, 0u);
            }
        }
        if (cSize >= n - ((n >> 6) + 2)) cSize = 0;                      // ZSTD_minGain, zstd_compress.c:3055-3057
    }
    if ((u64)op + 3 + (cSize ? cSize : n) > cap) return 0;               // dstSize_tooSmall
    if (cSize == 0) {                                                    // ZSTD_noCompressBlock
        const u32 h = 1 + (0 << 1) + (n << 3);
        if (lane == 0) { dst[op] = (u8)h; dst[op + 1] = (u8)(h >> 8); dst[op + 2] = (u8)(h >> 16); }
        wave_copy(dst + op + 3, src, n, lane);
        return (int)(op + 3 + n);
    }
    const u32 h = 1 + (2 << 1) + (cSize << 3);
    if (lane == 0) { dst[op] = (u8)h; dst[op + 1] = (u8)(h >> 8); dst[op + 2] = (u8)(h >> 16); }
    wave_fence();
#ifdef ZSTD_PROF
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long tc0 = __builtin_amdgcn_s_memtime();
#endif
    wave_copy(dst + op + 3, ws->tmp, cSize, lane);
#ifdef ZSTD_PROF
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (lane == 0) atomicAdd(&g_zstd_eprof[11], __builtin_amdgcn_s_memtime() - tc0);
#endif
    return (int)(op + 3 + cSize);
}



// ================================================================================================
// Frames of any size (inputs above 128 KiB): ZSTD_compress_frameChunk (zstd_compress.c:4504-4583) cuts the input into blocks
// of min(128 KiB, window) bytes that share the hash table, the repeat offsets and the Huffman table of the last block
// written compressed.  Blocks depend on each other in order, so a frame is one wave's work from the first block to the last;
// frames of a batch run side by side.  "fast" strategy only (levels 1-2 where the level table says so, or explicit
// parameters through ZSTD_compress_advanced).
//
// The match finder is the same exact speculation as above with what a later block needs: positions are offsets into the
// frame (table entries position + 2, ZSTD_WINDOW_START_INDEX), the table has 2^hashLog u32 entries in the HBM workspace and
// persists, candidates below the prefix start (ZSTD_getLowestPrefixIndex, zstd_compress_internal.h:1336-1347) are ignored,
// matches stop at the block's end, and the repeat offsets come in and go out (zstd_fast.c:481-489, :614-634).
// ================================================================================================
constexpr u32 kFrScratch = 1024;                // u16 slots for the slot-reuse rounds


template <int MLS>
__device__ u32 fast_frame_block(const gbyte* __restrict__ src, const u32 bs, const u32 be, const u32 hlog, u32* tab, lbyte* scr,
                                const FrameWindow& win, u32& rep1io, u32& rep2io, Workspace* ws_, u32* litSizeOut, int lane, const u32* sched_ = nullptr)
{
    GWorkspace* ws = (GWorkspace*)ws_;
    const u32* __restrict__ sched = uni_ptr(sched_);
    typedef __attribute__((address_space(1))) u32 gu32;
    typedef __attribute__((address_space(3))) uint16_t lu16;
    gu32* T = (gu32*)tab;
    lu16* scratch = (lu16*)scr;
    auto tget = [&](u32 h) -> u32 { return __hip_atomic_load((u32*)(T + h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto tput = [&](u32 h, u32 v) { __hip_atomic_store((u32*)(T + h), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    u32 q_ll = 0, q_off = 0, q_ml = 0, q_src = 0;           // sequence records parked one per lane, written 64 at a time
    auto flush_seq = [&](u32 first, u32 count) {
        if ((u32)lane < count) { ws->seqLL[first + lane] = q_ll; ws->seqOff[first + lane] = q_off; ws->seqML[first + lane] = q_ml; ws->seqSrc[first + lane] = q_src; }
    };
    auto put_seq = [&](u32 idx, u32 ll, u32 off, u32 ml, u32 srcpos) {
        const bool me = (u32)lane == (idx & 63u);
        q_ll = me ? ll : q_ll; q_off = me ? off : q_off; q_ml = me ? ml : q_ml; q_src = me ? srcpos : q_src;
        if ((idx & 63u) == 63u) flush_seq(idx - 63u, 64);
    };
    const u32 n = be;                                                    // iend
    const u32 prefixIdx = lowest_prefix(win, be + 2), prefixPos = prefixIdx - 2;   // :443-446
    u32 nbSeq = 0, litSize = 0, anchor = bs, s = bs + (bs == prefixPos ? 1u : 0u);
    u32 rep1 = rep1io, rep2 = rep2io, saved1 = 0, saved2 = 0;
    {   const u32 curr = s + 2, maxRep = curr - lowest_prefix(win, curr);           // :481-489
        if (rep2 > maxRep) { saved2 = rep2; rep2 = 0; }
        if (rep1 > maxRep) { saved1 = rep1; rep1 = 0; }
    }
    if (n >= 20) {
        const u32 ilimit = n - 16;                                       // iend - HASH_READ_SIZE - PREFETCH_SAFETY (:448)
        auto rep_loop = [&](u32 ip0) -> u32 {                            // immediate rep2 matches behind a sequence (:690-702)
            bool hit = uni(ld32(src + ip0) == ld32(src + ip0 - rep2) ? 1u : 0u) != 0;
            while (hit) {
                const u32 rLength = wave_count(src, ip0 + 4, ip0 + 4 - rep2, n, lane) + 4;
                const u32 t = rep2; rep2 = rep1; rep1 = t;
                const u32 hc = z_hash<MLS>(ld64(src + ip0), hlog);
                if (lane == 0) tput(hc, ip0 + 2);
                put_seq(nbSeq, 0, 1, rLength, ip0);
                wave_fence();
                nbSeq++;
                ip0 += rLength;
                hit = ip0 <= ilimit && uni(ld32(src + ip0) == ld32(src + ip0 - rep2) ? 1u : 0u) != 0;
            }
            return ip0;
        };
        for (;;) {                                                      // _start
            u32 i0 = 0; int ev = 0;                                     // 1 = table match, 2 = repcode
            u32 e_pos = 0, e_cand = 0, e_sig = 0, e_st = 0, e_half = 0, e_p1 = 0, e_h1 = 0, e_h2 = 0;
            bool e_ok1 = false, e_ok2 = false;
            for (;;) {
                const u32 K = i0 + (u32)lane, m = K >> 1, half = K & 1;
                u32 am, sg, st;
                fast_sched_at(sched, m, am, sg, st);
                const bool valid = (u64)s + am + sg + 1 < ilimit;       // ip3 < ilimit (:503, :631)
                const u32 pos = s + am + half;
                u64 seq = 0;
                if (valid) seq = ld64(src + pos);
                const u32 h = z_hash<MLS>(seq, hlog);
                u32 ent = 0;
                if (valid) ent = tget(h);
                bool rep_hit = false;
                if (valid && half == 0 && rep1 > 0) {                   // repcode at ip2 (:519-542)
                    const u32 r = pos + sg;
                    rep_hit = ld32(src + r) == ld32(src + r - rep1);
                }
                u32 m4 = ~(u32)seq;
                if (valid && ent >= prefixIdx) m4 = ld32(src + ent - 2);
                wave_fence();
                const u32 w = lds_first_lane_per_slot((u32)(uintptr_t)(scratch + (h & (kFrScratch - 1))), (u32)lane, valid);
                wave_fence();
                int f = 64;                                             // first lane whose slot an earlier lane of the batch names
                for (u64 d = ballot(valid && w != (u32)lane); d; d &= d - 1) {
                    const int j = (int)__builtin_ctzll(d);
                    if ((ballot(valid && h == lane_get(h, j)) & ((1ull << j) - 1)) != 0) { f = j; break; }
                }
                const int e = first_lane(ballot(!valid));
                const int lim = f < e ? f : e;
                const bool mat_hit = m4 == (u32)seq;
                const int el = first_lane(ballot(lane < lim && (rep_hit || mat_hit)));
                const int ncommit = (el < 64) ? el + 1 : lim;
                if (lane < ncommit) tput(h, pos + 2);
                wave_fence();
                if (el < 64) {
                    ev = lane_get((u32)rep_hit, el) ? 2 : 1;
                    e_pos = lane_get(pos, el); e_cand = lane_get(ent, el);
                    e_sig = lane_get(sg, el); e_st = lane_get(st, el); e_half = lane_get(half, el);
                    e_ok1 = el + 1 < 64 && lane_get((u32)valid, (el + 1) & 63) != 0;
                    e_p1 = lane_get(pos, (el + 1) & 63); e_h1 = lane_get(h, (el + 1) & 63);
                    e_ok2 = el + 2 < 64 && lane_get((u32)valid, (el + 2) & 63) != 0 && lane_get(pos, (el + 2) & 63) == e_pos + 2;
                    e_h2 = lane_get(h, (el + 2) & 63);
                    break;
                }
                if (e < 64 && e <= f) break;
                i0 += (u32)lim;
            }
            if (!ev) break;                                             // _cleanup
            {   // table write of ip1 that accompanies the hit (:537, :557, :586-598)
                const u32 ip1 = e_half == 0 ? e_pos + 1 : e_pos - 1 + e_sig;
                if (e_half == 0 || e_st <= 4) {
                    const u32 h1 = (e_ok1 && e_p1 == ip1) ? e_h1 : z_hash<MLS>(ld64(src + ip1), hlog);
                    if (lane == 0) tput(h1, ip1 + 2);
                    wave_fence();
                }
            }
            const u32 current0 = e_pos;
            u32 ip0, match0, offcode, mLength, maxback;
            if (ev == 2) { ip0 = e_pos + e_sig; match0 = ip0 - rep1; offcode = 1; maxback = 1; }       // :519-542
            else {                                                      // _offset (:651-665)
                ip0 = e_pos; match0 = e_cand - 2;
                rep2 = rep1; rep1 = ip0 - match0; offcode = rep1 + 3;
                const u32 a = ip0 - anchor, b = match0 - prefixPos; maxback = a < b ? a : b;
            }
            {   // catch-up and forward extension: the first round of both compares (64 bytes back, 256 bytes forward) is
                // issued together, one global round trip instead of two; longer runs continue with the loops
                const u32 fa = ip0 + 4, fb = match0 + 4;
                bool eqb = false;
                if ((u32)lane < maxback) eqb = src[ip0 - 1 - (u32)lane] == src[match0 - 1 - (u32)lane];
                const u32 o = 4u * (u32)lane, pa = fa + o;
                u32 x = 0xFFFFFFFFu;
                if (pa + 4 <= n) x = ld32(src + pa) ^ ld32(src + fb + o);
                else if (pa < n) {
                    x = 0;
                    for (u32 t = 0; t < n - pa; ++t) x |= (u32)(src[pa + t] ^ src[fb + o + t]) << (8 * t);
                    x |= 0xFFFFFFFFu << (8 * (n - pa));
                }
                u32 back = (u32)first_lane(ballot(!eqb));               // <= maxback for maxback <= 64
                if (back == 64) {
                    while (back < maxback) {
                        const u32 k = back + (u32)lane;
                        bool eq = false;
                        if (k < maxback) eq = src[ip0 - 1 - k] == src[match0 - 1 - k];
                        const int c = first_lane(ballot(!eq));
                        back += (u32)c;
                        if (c < 64) break;
                    }
                }
                if (back > maxback) back = maxback;
                u32 fwd;
                const int l = first_lane(ballot(x != 0));
                if (l == 64) fwd = 256 + wave_count(src, fa + 256, fb + 256, n, lane);
                else fwd = 4u * (u32)l + ((u32)__builtin_ctz(lane_get(x, l)) >> 3);
                ip0 -= back; match0 -= back; mLength = 4 + back + fwd;   // _match (:669)
            }
            {
                const u32 ll = ip0 - anchor;
                put_seq(nbSeq, ll, offcode, mLength, anchor);
                litSize += ll; nbSeq++;
            }
            ip0 += mLength; anchor = ip0;
            if (ip0 <= ilimit) {                                        // :678-702
                const u32 ha = e_ok2 ? e_h2 : z_hash<MLS>(ld64(src + current0 + 2), hlog), hb2 = z_hash<MLS>(ld64(src + ip0 - 2), hlog);
                if (lane == 0) { tput(ha, current0 + 2 + 2); tput(hb2, ip0 - 2 + 2); }
                wave_fence();
                if (rep2 > 0) { ip0 = rep_loop(ip0); anchor = ip0; }
            }
            s = ip0;
        }
    }
    saved2 = (saved1 != 0 && rep1 != 0) ? saved1 : saved2;               // :614-630
    rep1io = rep1 ? rep1 : saved1;
    rep2io = rep2 ? rep2 : saved2;
    flush_seq(nbSeq & ~63u, nbSeq & 63u);
    wave_fence();
    *litSizeOut = litSize + gather_literals(src, n, ws, nbSeq, anchor, lane);
    return nbSeq;
}

// The fast strategy has a second finder for lone frames and small batches: fast_block<MLS, u32, false, true>, the chunk kernels'
// finder (hash table AND source ring in LDS, chain batches that resolve several sequences per batch, zstd_chain.h) on offsets
// into the frame - 2^hashLog u32 entries, so hashLog <= 14 (64 KB of LDS: one frame per CU); fast_frame_block above (table in
// HBM, 9 frames per CU) is the bulk variant.  Same output, both driven by the same tests.
struct FrameLds {
    union {                                      // never live at the same time: a block is parsed, then coded
        EntropyLds e;
        uint16_t scratch[2 * kDfScratch];        // slot-reuse rounds of the match finder: one table (fast) or two (double-fast)
    };
    HufCT prev;                                  // Huffman table of the last confirmed block
};

// One frame: returns its size, 0 when it does not fit `cap` (dstSize_tooSmall) or the parameters have no GPU parse.
// LT: the fast finder with its table in LDS (`tab` = 2^hlog u32 entries followed by the 512-byte source ring, both LDS);
// otherwise `tab` is this workgroup's slice of the HBM workspace.
template <bool LT>
__device__ long long zstd_frame_wave(const u8* __restrict__ src, const u64 n64, u8* __restrict__ dst, const u64 cap, const u32 strategy,
                                     const u32 windowLog, const u32 hlog, const u32 clog, const u32 mls, const u32 targetLength,
                                     FrameLds* L, Workspace* ws, u32* tab, int lane, const u32* sched = nullptr)
{
    if (n64 >= (1ull << 31) || windowLog < 10 || windowLog > 31 || hlog < 6 || mls < 4 || mls > 7) return 0;
    if (strategy == 1 ? (targetLength > (1u << 17) || (targetLength > 1 && sched == nullptr)) : (strategy != 2 || clog < 6)) return 0;
    const u32 n = (u32)n64;
    const u32 blockMax = windowLog < 17 ? 1u << windowLog : kBlockMax;   // MIN(ZSTD_BLOCKSIZE_MAX, windowSize), zstd_compress.c:2129-2131
    const u32 single = ((1ull << windowLog) >= n64) ? 1u : 0u;
    const u32 fcs = (n >= 256) + (n >= 65536 + 256);
    u64 op = 5 + (single ? 0 : 1) + (fcs == 0 ? single : (fcs == 1 ? 2 : 4));
    if (cap < op + 3) return 0;
    if (lane == 0) {                                                     // ZSTD_writeFrameHeader :4586-4632
        u32 p = 5;
        dst[0] = 0x28; dst[1] = 0xB5; dst[2] = 0x2F; dst[3] = 0xFD;
        dst[4] = (u8)((single << 5) | (fcs << 6));
        if (!single) dst[p++] = (u8)((windowLog - 10) << 3);
        if (fcs == 0) { if (single) dst[p] = (u8)n; }
        else if (fcs == 1) { const u32 v = n - 256; dst[p] = (u8)v; dst[p + 1] = (u8)(v >> 8); }
        else st32(dst + p, n);
    }
    if (n == 0) { if (lane == 0) { dst[op] = 1; dst[op + 1] = 0; dst[op + 2] = 0; } return (long long)(op + 3); }   // ZSTD_writeEpilogue
    if constexpr (LT) {                                                  // fresh table (ZSTD_reset_matchState clears it for a new frame)
        if (strategy != 1) return 0;
        typedef __attribute__((address_space(3))) u32 L32;
        for (u32 i = (u32)lane; i < (1u << hlog); i += kWave) ((L32*)tab)[i] = 0;
        wave_fence();
    } else {
        typedef u32 u4 __attribute__((ext_vector_type(4)));
        const u32 quads = ((1u << hlog) + (strategy == 2 ? 1u << clog : 0u)) / 4;
        for (u32 i = (u32)lane; i < quads; i += kWave) ((__attribute__((address_space(1))) u4*)tab)[i] = u4{0, 0, 0, 0};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    }
    FrameWindow win = { 2u, 2u, 1u << windowLog };
    FrameHuf fh = { (ZLDS HufCT*)&L->prev, 0u, 0u, strategy == 1 ? targetLength : 0u };
    u32 rep1 = 1, rep2 = 4;
    for (u32 bs = 0; bs < n; bs += blockMax) {
        const u32 bn = n - bs < blockMax ? n - bs : blockMax;
        const u32 last = bs + bn == n ? 1u : 0u;
        if (bs + 2 > win.maxDist) {                                      // ZSTD_window_enforceMaxDist at the block's start (:4532)
            const u32 low = bs + 2 - win.maxDist;
            if (win.lowLimit < low) win.lowLimit = low;
            if (win.dictLimit < win.lowLimit) win.dictLimit = win.lowLimit;
        }
        u32 cSize = 0;
        if (bn >= 7) {                                                   // ZSTD_buildSeqStore :3268
            u32 litSize = 0, nbSeq, nr1 = rep1, nr2 = rep2;
            const gbyte* g = (const gbyte*)src;
            if constexpr (LT) {
                lbyte* lt = (lbyte*)tab;
                if (mls == 4) nbSeq = fast_block<4, u32, false, true>(g, bs + bn, hlog, lt, ws, &litSize, lane, 0u, bs, &win, &nr1, &nr2, sched);
                else if (mls == 5) nbSeq = fast_block<5, u32, false, true>(g, bs + bn, hlog, lt, ws, &litSize, lane, 0u, bs, &win, &nr1, &nr2, sched);
                else if (mls == 6) nbSeq = fast_block<6, u32, false, true>(g, bs + bn, hlog, lt, ws, &litSize, lane, 0u, bs, &win, &nr1, &nr2, sched);
                else nbSeq = fast_block<7, u32, false, true>(g, bs + bn, hlog, lt, ws, &litSize, lane, 0u, bs, &win, &nr1, &nr2, sched);
            } else {
                if (strategy == 2) {
                    lbyte* sc = (lbyte*)L->scratch;
                    if (mls == 4) nbSeq = dfast_block<4, true>(g, bs + bn, hlog, clog, tab, sc, ws, &litSize, lane, bs, &win, &nr1, &nr2);
                    else if (mls == 5) nbSeq = dfast_block<5, true>(g, bs + bn, hlog, clog, tab, sc, ws, &litSize, lane, bs, &win, &nr1, &nr2);
                    else if (mls == 6) nbSeq = dfast_block<6, true>(g, bs + bn, hlog, clog, tab, sc, ws, &litSize, lane, bs, &win, &nr1, &nr2);
                    else nbSeq = dfast_block<7, true>(g, bs + bn, hlog, clog, tab, sc, ws, &litSize, lane, bs, &win, &nr1, &nr2);
                }
                else if (mls == 4) nbSeq = fast_frame_block<4>(g, bs, bs + bn, hlog, tab, (lbyte*)L->scratch, win, nr1, nr2, ws, &litSize, lane, sched);
                else if (mls == 5) nbSeq = fast_frame_block<5>(g, bs, bs + bn, hlog, tab, (lbyte*)L->scratch, win, nr1, nr2, ws, &litSize, lane, sched);
                else if (mls == 6) nbSeq = fast_frame_block<6>(g, bs, bs + bn, hlog, tab, (lbyte*)L->scratch, win, nr1, nr2, ws, &litSize, lane, sched);
                else nbSeq = fast_frame_block<7>(g, bs, bs + bn, hlog, tab, (lbyte*)L->scratch, win, nr1, nr2, ws, &litSize, lane, sched);
            }
            wave_fence();
            cSize = entropy_compress<true>(ws->tmp, ws, litSize, nbSeq, &L->e, lane, strategy, &fh);
            if (cSize >= bn - ((bn >> 6) + 2)) cSize = 0;                // ZSTD_minGain, zstd_compress.c:3055-3057
            if (bs != 0 && cSize < 25) {                                 // ZSTD_isRLE, never for the first block (:4372-4383)
                const u32 b0 = src[bs];
                bool same = true;
                for (u32 i = (u32)lane; i < bn && same; i += kWave) same = src[bs + i] == b0;
                if (!ballot(!same)) { cSize = 1; if (lane == 0) ws->tmp[0] = (u8)b0; }
            }
            if (cSize > 1) {                                             // ZSTD_blockState_confirmRepcodesAndEntropyTables
                rep1 = nr1; rep2 = nr2;
                if (fh.fresh) {
                    for (u32 i = (u32)lane; i < 256; i += kWave) { L->prev.nbBits[i] = L->e.huf.nbBits[i]; L->prev.val[i] = L->e.huf.val[i]; }
                    if (lane == 0) L->prev.tableLog = L->e.huf.tableLog;
                    fh.repeat = 1;
                }
            }
            wave_fence();
        }
        const u32 body = cSize == 0 ? bn : cSize;
        if (op + 3 + body > cap) return 0;                               // dstSize_tooSmall
        const u32 h = last + ((cSize == 0 ? 0u : (cSize == 1 ? 1u : 2u)) << 1) + ((cSize > 1 ? cSize : bn) << 3);
        if (lane == 0) { dst[op] = (u8)h; dst[op + 1] = (u8)(h >> 8); dst[op + 2] = (u8)(h >> 16); }
        if (cSize == 0) wave_copy(dst + op + 3, src + bs, bn, lane);     // ZSTD_noCompressBlock
        else wave_copy(dst + op + 3, ws->tmp, cSize, lane);
        wave_fence();
        op += 3 + body;
    }
    return (long long)op;
}


// ================================================================================================
// Decoder: ZSTD_decompress (algos/zstd/lib/decompress/zstd_decompress.c:1252-1389, :1005-1111;
// zstd_decompress_block.c:162-364, :497-707, :1339-1460, :1007-1210, :2444-2552; huf_decompress.c;
// fse_decompress.c; entropy_common.c:40-200, 234-310).  Any malformed input yields -1.
// ================================================================================================
struct BitR {                       // backward bit reader (bitstream.h:262-420); never loads outside [p, p+n)
    const u8* p; int pos; u64 win; int wbase;
    __device__ bool init(const u8* s, u32 len)
    {
        if (!len) return false;
        const u32 last = s[len - 1];
        if (!last) return false;
        p = s; pos = (int)(len - 1) * 8 + hb(last); wbase = 0x7FFFFFFF; win = 0;
        return true;
    }
    __device__ void refill()
    {
        const int q = ((pos + 7) >> 3) - 8;
        if (q >= 0) { win = ld64(p + q); wbase = q * 8; }
        else {
            const int cnt = (pos + 7) >> 3;
            win = 0;
            for (int i = 0; i < cnt; ++i) win |= (u64)p[i] << (8 * i);
            wbase = 0;
        }
    }
    __device__ u32 peek(int nb)     // bits [pos-nb, pos), zero-padded below bit 0; nb <= 32
    {
        if (nb == 0 || pos <= 0) return 0;
        if (pos - nb < wbase || pos > wbase + 64) refill();
        const u64 mask = (1ull << nb) - 1;
        if (pos - nb >= wbase) return (u32)((win >> (pos - nb - wbase)) & mask);
        return (u32)((win << (nb - pos)) & mask);               // wbase == 0 and pos < nb
    }
    __device__ u32 read(int nb) { const u32 v = peek(nb); pos -= nb; return v; }
};

// Backward bit reader for the serial decode loops (one lane per stream).  The stream is staged through a window in
// LDS that the whole wavefront refills with coalesced loads between rounds of the serial loop (bytes outside the
// stream are stored as 0), so the loop itself never touches global memory: `grab` is one 8-byte LDS read that
// left-aligns the next >= 57 unread bits in a container (BIT_reloadDStream), `take` consumes fields from its top with
// three instructions (BIT_readBitsFast).
struct BitL {
    int pos;                              // bits of the stream still unread (may go negative: over-read)
    int wb;                               // stream byte offset of window byte 0
    u64 c;                                // container: unread bits, next one at bit 63
    __device__ __forceinline__ void grab(const lbyte* win)
    {
        if (pos <= 0) { c = 0; return; }
        const int e = (pos + 7) >> 3;      // the 8 bytes that end with the byte holding bit pos-1
        c = lds_ld64(win + (e - 8 - wb)) << (8 * e - pos);
    }
    __device__ __forceinline__ u32 take(u32 nb)   // nb <= 32 bits from the top of the container; nb == 0 -> 0
    {
        const u32 v = (u32)((c >> 1) >> (63 - nb));
        c <<= nb; pos -= (int)nb;
        return v;
    }
    __device__ __forceinline__ u32 top(u32 nb) const { return (u32)((c >> 1) >> (63 - nb)); }
    __device__ __forceinline__ void skip(u32 nb) { c <<= nb; pos -= (int)nb; }
};
// position of the last set bit of a stream's final byte: unread bit count, or -1 for an invalid stream
__device__ __forceinline__ int bit_stream_bits(const u8* s, u32 len)
{
    if (!len) return -1;
    const u32 last = s[len - 1];
    if (!last) return -1;
    return (int)(len - 1) * 8 + hb(last);
}
// 16 stream bytes at offset `off` (bytes outside [0, n) read as 0)
__device__ __forceinline__ u32x4 stream16(const gbyte* p, u32 n, int off)
{
    if (off >= 0 && (u32)off + 16 <= n) { typedef u32x4 __attribute__((aligned(1))) U; return *(const __attribute__((address_space(1))) U*)(p + off); }
    u32 w[4] = {0, 0, 0, 0};
    for (int i = 0; i < 16; ++i) { const int a = off + i; if (a >= 0 && (u32)a < n) w[i >> 2] |= (u32)p[a] << (8 * (i & 3)); }
    return u32x4{w[0], w[1], w[2], w[3]};
}

struct FseD { u16 newState; u8 symbol; u8 nbBits; };

struct DecLds {                     // 7.3 KB = 6 LDS granules of 1280 bytes (round 5; 12.4 KB = 10 granules before): the serial entropy
                                    // loops are latency chains and throughput follows residency - 6 / 9 / 12 blocks per CU decode the
                                    // 8 GiB bench in 108.7 / 78.5 / 63.7 ms - so the Huffman table and the three FSE tables, which are
                                    // never read in the same phase of a block, share their LDS.  What a LATER block of the frame may
                                    // reuse (treeless literals, repeat-mode sequence tables) is parked in the block's global workspace
                                    // (kDecHufPark / kDecFsePark) by the blocks that are not the frame's last, and fetched back by the
                                    // block that asks for it; a frame of one block - every chunk of a batch - never does either.
    union {
        u16 huf[2048];              // literals: symbol | nbBits << 8, table logs <= 11; a 12-bit table (accepted by the reference,
                                    // never produced by zstd) lives in the block's global workspace instead
        // Sequences: three-byte cells - new state base (9 bits) | additional bits of the code (5) | code (6) | state bits (4) -
        // so that the three tables take 3840 bytes, not 5120, and a block five LDS granules, not six (24 blocks per CU instead
        // of 21).  A table is built with four-byte cells where it will stand and packed in place (seq_table); building one may
        // run over the tables behind it, which are built or fetched after it, and the last one runs into seqBuild.
        struct { u8 ll3[1536], of3[768], ml3[1536], seqBuild[512]; };
        struct { s16 wnorm[256]; u16 wnext[256]; };           // reading the FSE table of the Huffman weights (before huf is built)
    };
    // 1920 bytes that three phases of a block use one after the other (6336 bytes in all = 5 granules):
    union {
        u8 hwin[4][480] __attribute__((aligned(16)));       // literals by one lane per stream (12-bit tables, single streams): windows of
                                                             // the four Huffman streams (256 symbols need <= 352 bytes + the 8-byte container)
        struct {                                             // reading a table description (Huffman weights before the streams, FSE tables after them)
            s16 norm[64];                                    // (sequence tables: codes <= 52; the weights' table may name 256 symbols: wnorm)
            u8 weights[256];
            FseD wdt[64];
            u8 scratch[768];                                 // staged description / occupancy + next-state scratch of the table builders
        };
        struct {                                             // decoding and executing the sequences (filled after the tables are built)
            u8 bswin[768];                                   // window of the sequence bitstream (one tile of 64 sequences needs <= 712 bytes + the 8-byte container)
            u32 seqLL[64], seqML[64], seqOff[64];
            u32 llTab[36], mlTab[53];                        // base | additional bits << 24 per length code
        };
    };
    // (behind the unions: seqBuild, bswin and the seq arrays - 2048 bytes in a row, all of them dead while a tile of sequences is
    //  executed - hold the tile's output block then: lz_group.hip.h, 64 bytes of history + kDecGrpBlock + 16)
    u32 llLog, ofLog, mlLog, hufLog;
    u32 haveLL, haveOF, haveML, haveHuf;
    u32 rep[3];
    u32 zeroWord, dumpWord, pad_[2];                          // a cell of zeros / a sink for the idle lanes of the sequence loop
#ifdef ZSTD_DEC_LDS_PAD                                      /* residency experiments: fewer blocks per CU */
    u8 residencyPad[ZSTD_DEC_LDS_PAD];
#endif
};
constexpr u32 kDecGrpBlock = 1920;
static_assert(offsetof(DecLds, bswin) == offsetof(DecLds, seqBuild) + 512 && offsetof(DecLds, llTab) >= offsetof(DecLds, seqBuild) + kGrpHist + kDecGrpBlock + 16,
              "the output block of a tile runs from seqBuild to the end of the seq arrays");
static_assert(sizeof(DecLds) <= 6400 || sizeof(DecLds) > 20000, "five LDS granules per block");

// FSE_readNCount (entropy_common.c:40-186), lane-serial; bytes consumed or 0xFFFFFFFF
__device__ u32 fse_read_ncount(s16* norm, u32* maxSym, u32* tableLog, const u8* src, u32 n, u32 maxLog)
{
    u32 bp = 0, charnum = 0, nbBits; int previous0 = 0, remaining, threshold;
    const u32 maxSV1 = *maxSym + 1;
    auto peek32 = [&](u32 bitpos) -> u32 {
        const u32 by = bitpos >> 3;
        u64 w = 0;
        for (u32 i = 0; i < 5; ++i) if (by + i < n) w |= (u64)src[by + i] << (8 * i);
        return (u32)(w >> (bitpos & 7));
    };
    for (u32 i = 0; i < maxSV1; ++i) norm[i] = 0;
    if (n < 1) return 0xFFFFFFFFu;
    nbBits = (peek32(0) & 0xF) + 5; bp = 4;
    if (nbBits > 15 || nbBits > maxLog) return 0xFFFFFFFFu;
    *tableLog = nbBits;
    remaining = (1 << nbBits) + 1; threshold = 1 << nbBits; nbBits++;
    for (;;) {
        if (previous0) {
            for (;;) {
                const u32 c = peek32(bp) & 3; bp += 2;
                if (c == 3) { charnum += 3; if (bp > 8 * n + 64) return 0xFFFFFFFFu; continue; }
                charnum += c; break;
            }
            if (charnum >= maxSV1) break;
        }
        {   const u32 bits = peek32(bp);
            const int max = (2 * threshold - 1) - remaining; int count;
            if ((bits & (u32)(threshold - 1)) < (u32)max) { count = (int)(bits & (u32)(threshold - 1)); bp += nbBits - 1; }
            else { count = (int)(bits & (u32)(2 * threshold - 1)); if (count >= threshold) count -= max; bp += nbBits; }
            count--;
            remaining -= count < 0 ? -count : count;
            norm[charnum++] = (s16)count;
            previous0 = !count;
            if (remaining < threshold) {
                if (remaining <= 1) break;
                nbBits = (u32)hb((u32)remaining) + 1; threshold = 1 << (nbBits - 1);
            }
            if (charnum >= maxSV1) break;
        }
    }
    if (remaining != 1 || charnum > maxSV1) return 0xFFFFFFFFu;
    *maxSym = charnum - 1;
    const u32 used = (bp + 7) >> 3;
    if (used > n) return 0xFFFFFFFFu;
    return used;
}
// FSE decoding table (ZSTD_buildFSETable, zstd_decompress_block.c:497-615), one symbol per lane: the spread is the encoder's
// (fse_spread_wave); the j-th state of a symbol (in increasing state order) continues from value norm + j, which gives its
// bit count and base.  norm = this lane's normalised count.  false when the counts do not fill the table.
// occ: 2^tableLog bytes, next: 64 u16 of LDS scratch (the decoder lends its bitstream window, not yet in use)
__device__ bool fse_build_dtable_wave(FseD* dt, const int norm, u32 maxSym, u32 tableLog, u8* occ, u16* next, int lane)
{
    const u32 size = 1u << tableLog;
    const bool valid = (u32)lane <= maxSym;
    const u32 w = !valid ? 0u : (norm == -1 ? 1u : (norm > 0 ? (u32)norm : 0u));
    if (wsum(w) != size) return false;                     // (the reference notices the same through pos != 0 after its walk)
    fse_spread_wave(occ, norm, maxSym, tableLog, lane, [&](u32 state, u32 sym) { dt[state].symbol = (u8)sym; });
    if (valid) next[lane] = (u16)w;
    wave_fence();
    for (u32 u0 = 0; u0 < size; u0 += kWave) {
        const u32 u = u0 + (u32)lane;
        const bool in = u < size;
        const u32 sym = in ? (u32)dt[u].symbol : 0u;
        const u64 same = same_value6(sym, in);
        if (in) {
            const u32 base = next[sym], ns = base + below(same);
            const u32 nb = tableLog - (u32)hb(ns);
            dt[u].nbBits = (u8)nb; dt[u].newState = (u16)((ns << nb) - size);
            if (lane == top_lane(same)) next[sym] = (u16)(base + (u32)__builtin_popcountll(same));
        }
        wave_fence();
    }
    return true;
}
// (the table of the Huffman weights, <= 64 states of <= 13 symbols, is built by the lane that also decodes them)
__device__ bool fse_build_dtable(FseD* dt, const s16* norm, u32 maxSym, u32 tableLog, u16* next)
{
    const u32 size = 1u << tableLog, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    u32 high = size - 1, pos = 0;
    for (u32 s = 0; s <= maxSym; ++s) {
        if (norm[s] == -1) { dt[high--].symbol = (u8)s; next[s] = 1; }
        else next[s] = (u16)norm[s];
    }
    for (u32 s = 0; s <= maxSym; ++s)
        for (int i = 0; i < norm[s]; ++i) {
            dt[pos].symbol = (u8)s;
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    if (pos != 0) return false;
    for (u32 u = 0; u < size; ++u) {
        const u8 sym = dt[u].symbol; const u32 ns = next[sym]++;
        dt[u].nbBits = (u8)(tableLog - (u32)hb(ns));
        dt[u].newState = (u16)((ns << dt[u].nbBits) - size);
    }
    return true;
}

// HUF_readStats + X1 table (entropy_common.c:234-310, huf_decompress.c:380-470); all lanes call it; bytes used or 0xFFFFFFFF.
// What is serial by format stays on lane 0 - the FSE-coded weights (two interleaved states in one bitstream), read from a copy
// of the description (< 130 bytes) that the wave stages in LDS first, so the chain never waits on global memory.  Everything
// after the weights is the wave's: four symbols per lane (symbol = 64 k + lane), the weight histogram and each symbol's rank
// inside its weight class from ballots, the class starts from an unrolled scan over the 12 weights (no indexed local array:
// those live in scratch memory), the cells of short codes by whole-wave fills and those of long codes by their own lanes.
// (Inlined since round 5.  Round 2 had made this and seq_table calls because, inlined, they held the decoder at 2 waves per SIMD;
//  but values that live across a call sit in callee-saved VGPRs, which are numbered sparsely up to v143, and the kernel then
//  counts 147 VGPRs whatever amdgpu_waves_per_eu asks for.  Inlined, the budget of 6 waves per SIMD is met with 85 VGPRs and
//  69 dwords of scratch, and the decoder's residency is what its LDS allows: DESIGN.md section 4.)
__device__ __forceinline__ u32 huf_read_table(DecLds* L, const u8* src, u32 n, u16* big, int lane)
{
    u8* w = L->weights;
    if (!n) return 0xFFFFFFFFu;
    const u32 h0 = uni(src[0]);
    u32 nsym = 0, used = 0;
    if (h0 >= 128) {
        nsym = h0 - 127; used = 1 + (nsym + 1) / 2;
        if (used > n) return 0xFFFFFFFFu;
        for (u32 b = (u32)lane; b < (nsym + 1) / 2; b += kWave) { const u32 v = src[1 + b]; w[2 * b] = (u8)(v >> 4); w[2 * b + 1] = (u8)(v & 15); }
    } else {
        const u32 isz = h0;
        used = 1 + isz;
        if (used > n || isz < 2) return 0xFFFFFFFFu;
        u8* stg = L->scratch;
        for (u32 b = (u32)lane; b < isz; b += kWave) stg[b] = src[1 + b];
        wave_fence();
        u32 res = 0;
        if (lane == 0) {
            u32 maxSym = 255, tl; BitR b;
            res = 0xFFFFFFFFu;
            const u32 h = fse_read_ncount(L->wnorm, &maxSym, &tl, stg, isz, 6);
            if (h != 0xFFFFFFFFu && h < isz && fse_build_dtable(L->wdt, L->wnorm, maxSym, tl, L->wnext) && b.init(stg + h, isz - h)) {
                u32 s1 = b.read((int)tl), s2 = b.read((int)tl), k = 0;
                for (;;) {                                                   // FSE_decompress_usingDTable tail (fse_decompress.c:220-245)
                    if (k > 253) { k = 0xFFFFFFFFu; break; }
                    w[k++] = L->wdt[s1].symbol; s1 = L->wdt[s1].newState + b.read(L->wdt[s1].nbBits);
                    if (b.pos < 0) { w[k++] = L->wdt[s2].symbol; break; }
                    if (k > 253) { k = 0xFFFFFFFFu; break; }
                    w[k++] = L->wdt[s2].symbol; s2 = L->wdt[s2].newState + b.read(L->wdt[s2].nbBits);
                    if (b.pos < 0) { w[k++] = L->wdt[s1].symbol; break; }
                }
                res = k;
            }
        }
        wave_fence();
        nsym = uni(res);
        if (nsym == 0xFFFFFFFFu) return 0xFFFFFFFFu;
    }
    wave_fence();
    // ---- the wave's part: symbol 64 k + lane in slot k ----
    u32 wk[4]; u32 part = 0; bool bad = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u32 i = 64u * k + (u32)lane;
        wk[k] = i < nsym ? (u32)w[i] : 0u;
        bad = bad || wk[k] > 12;
        part += (1u << (wk[k] & 15)) >> 1;
    }
    if (ballot(bad) != 0) return 0xFFFFFFFFu;
    const u32 total = wsum(part);
    if (total == 0) return 0xFFFFFFFFu;
    const u32 tableLog = (u32)hb(total) + 1;
    if (tableLog > 12) return 0xFFFFFFFFu;
    {   // the last symbol's weight is implied: what is left must be a power of two (entropy_common.c:290-300)
        const u32 rest = (1u << tableLog) - total;
        if ((1u << hb(rest)) != rest) return 0xFFFFFFFFu;
        const u32 lastW = (u32)hb(rest) + 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (64u * k + (u32)lane == nsym) wk[k] = lastW;
        nsym++;
    }
    u32 pos[4] = {0, 0, 0, 0};
    u32 cur = 0;
#pragma unroll
    for (u32 wv = 1; wv <= 12; ++wv) {                                       // class starts: weight ascending, symbols ascending inside
        const u64 m0 = ballot(wk[0] == wv), m1 = ballot(wk[1] == wv), m2 = ballot(wk[2] == wv), m3 = ballot(wk[3] == wv);
        const u32 c0 = (u32)__builtin_popcountll(m0), c1 = (u32)__builtin_popcountll(m1), c2 = (u32)__builtin_popcountll(m2), c3 = (u32)__builtin_popcountll(m3);
        const u32 sh = wv - 1;
        if (wk[0] == wv) pos[0] = cur + (below(m0) << sh);
        if (wk[1] == wv) pos[1] = cur + ((c0 + below(m1)) << sh);
        if (wk[2] == wv) pos[2] = cur + ((c0 + c1 + below(m2)) << sh);
        if (wk[3] == wv) pos[3] = cur + ((c0 + c1 + c2 + below(m3)) << sh);
        const u32 rank = c0 + c1 + c2 + c3;
        if (wv == 1 && (rank < 2 || (rank & 1))) return 0xFFFFFFFFu;          // by construction there are at least two weight-1 symbols
        cur += rank << sh;
    }
    typedef __attribute__((address_space(3))) u16 LT;
    typedef __attribute__((address_space(1))) u16 GT;
    LT* lt = (LT*)L->huf; GT* gt = (GT*)big;
    const bool toBig = tableLog == 12;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u32 wv = wk[k], len = wv ? 1u << (wv - 1) : 0u;
        const u32 cell = (64u * k + (u32)lane) | ((tableLog + 1 - wv) << 8);
        if (len && len <= 8) {                                               // long codes: the symbol's own lane
            if (toBig) for (u32 t = 0; t < len; ++t) gt[pos[k] + t] = (u16)cell;
            else for (u32 t = 0; t < len; ++t) lt[pos[k] + t] = (u16)cell;
        }
        for (u64 m = ballot(len > 8); m; m &= m - 1) {                        // short codes (few symbols, many cells): the whole wave
            const int j = (int)__builtin_ctzll(m);
            const u32 p0 = lane_get(pos[k], j), ln = lane_get(len, j), cl = lane_get(cell, j);
            if (toBig) for (u32 t = (u32)lane; t < ln; t += kWave) gt[p0 + t] = (u16)cl;
            else for (u32 t = (u32)lane; t < ln; t += kWave) lt[p0 + t] = (u16)cl;
        }
    }
    if (lane == 0) L->hufLog = tableLog;
    wave_fence();
    return used;
}
// one Huffman stream, executed by ONE lane
// Lanes 0..3 (nstreams of them) decode one Huffman stream each: rounds of up to 256 symbols (HUF_decodeStreamX1, four
// symbols per container), the whole wave refilling the LDS windows in between.  All 64 lanes call it; returns false
// on a lane whose stream is malformed.  (Kept inline: as a called function under a partial exec mask this loop was
// miscompiled by ROCm 7.2's hipcc.)
// lenient: the stream is one of four that the reference decodes with its fast loops (huf_decompress.c :790-840 / :1600-1655), which
// make no end-of-stream check: bits left over are no error (the symbols are the same), and a last byte of zero has no end mark to
// find - all eight bits are data (HUF_initFastDStream :167-176).  A stream that runs out of bits is still refused.
template <class TAB>
__device__ __forceinline__ bool huf_decode_streams_t(DecLds* L, u8* out_, u32 nout, const u8* src_, u32 n, TAB tab, const u32 tl,
                                                     const bool mine, int lane, const bool lenient = false)
{
    gbyte* out = (gbyte*)out_; const gbyte* src = (const gbyte*)src_;
    BitL b; b.pos = mine ? bit_stream_bits(src_, n) : 0; b.wb = 0; b.c = 0;
    if (lenient && mine && n != 0 && src_[n - 1] == 0) b.pos = (int)(8u * n);
    bool ok = !mine || b.pos >= 0;
    u32 i = 0;
    const u32 todo = (mine && ok) ? nout : 0;
    while (ballot(i < todo) != 0) {
        {   // refill: 15 lanes x 32 bytes per stream, window = the 480 bytes that end with the byte holding bit pos-1
            const int j = lane >> 4, k = lane & 15;
            const int e = (__shfl(b.pos, j) + 7) >> 3;
            const u64 sp = ((u64)(u32)__shfl((int)(u32)((u64)src >> 32), j) << 32) | (u32)__shfl((int)(u32)(u64)src, j);
            const u32 sn = (u32)__shfl((int)n, j);
            const bool act = __shfl((int)(i < todo), j) != 0;
            if (act && k < 15) {
                const gbyte* p = (const gbyte*)sp;
                const int wb = e - 480;
                typedef __attribute__((address_space(3))) u32x4 LV;
                lbyte* w = (lbyte*)&L->hwin[j][0];
                *(LV*)(w + 32 * k) = stream16(p, sn, wb + 32 * k);
                *(LV*)(w + 32 * k + 16) = stream16(p, sn, wb + 32 * k + 16);
            }
            b.wb = ((b.pos + 7) >> 3) - 480;
            wave_fence();
        }
        if (i < todo) {
            const lbyte* win = (const lbyte*)&L->hwin[lane & 3][0];
            const u32 stop = i + 256 < todo ? i + 256 : todo;
            for (; i + 8 <= stop; i += 8) {               // 8 symbols per store
                u64 acc = 0;
                for (int g = 0; g < 2; ++g) {
                    b.grab(win);
                    for (int k = 0; k < 4; ++k) {
                        const u32 cell = tab[b.top(tl)];
                        acc |= (u64)(cell & 0xFF) << (8 * (4 * g + k)); b.skip(cell >> 8);
                    }
                }
                st64(out + i, acc);
            }
            if (stop == todo) for (; i < todo; ++i) {
                b.grab(win);
                const u32 cell = tab[b.top(tl)];
                out[i] = (u8)cell; b.skip(cell >> 8);
            }
        }
        wave_fence();
    }
    return ok && (!mine || b.pos == 0 || (lenient && b.pos > 0));
}
__device__ __forceinline__ bool huf_decode_streams(DecLds* L, u8* out, u32 nout, const u8* src, u32 n, const u16* big, const bool mine, int lane,
                                                   const bool lenient = false)
{
    typedef __attribute__((address_space(3))) const u16 LT;
    typedef __attribute__((address_space(1))) const u16 GT;
    const u32 tl = uni(L->hufLog);
    if (tl == 12) return huf_decode_streams_t(L, out, nout, src, n, (GT*)big, tl, mine, lane, lenient);
    return huf_decode_streams_t(L, out, nout, src, n, (LT*)L->huf, tl, mine, lane, lenient);
}

#ifdef ZSTD_PROF
__device__ unsigned long long g_zstd_dprof[16];
#define DSTAGE(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); dacc[k] += t_ - dprev; dprev = t_; } while (0)
#else
#define DSTAGE(k) do { } while (0)
#endif
// The four Huffman streams of a literals section, 16 lanes per stream (zstd_hufspec.h): every lane runs its segment from a
// guessed start - in the first round only the segment's tail, which is enough to tell where the lane leaves it (zstd_hufspec.h:
// kProbeBits) -, lanes whose predecessor left somewhere else run again until nothing changes (lane 0 of a stream is exact from the
// second round, so round r + 1 settles lane r; prefix codes fall into step within a few symbols, so a probe, one counting run and
// the writing run are the rule: ~2.1 passes over the streams, three before round 6), then the
// symbol counts become write positions by a prefix sum and one more run writes.  so / sl: this lane's stream inside src,
// oo / no: where its symbols go and how many there must be.  Accepts exactly what the one-lane-per-stream decoder accepts:
// `no` symbols that end on bit 0 of the stream.  Table logs <= 11 (table in LDS).
__device__ __forceinline__ bool huf_decode_spec(DecLds* L, u8* out_, const u8* src_, const u32 so, const u32 sl, const u32 oo, const u32 no,
                                                const u32 tl, int lane
#ifdef ZSTD_PROF
                                                , unsigned long long* dacc, unsigned long long& dprev
#endif
                                                )
{
    typedef __attribute__((address_space(3))) const u16 LT;
    DSTAGE(5);
    const gbyte* sp = (const gbyte*)src_ + so;
    gbyte* out = (gbyte*)out_;
    const int sub = lane & 15;
    int top = bit_stream_bits(src_ + so, sl);
    bool ok = top >= 0;
    if (!ok) top = 0;
    const int seg = hufspec::seg_bits(top, 16);
    const int segTop = hufspec::seg_start(top, seg, sub);
    const int limit = hufspec::seg_start(top, seg, sub + 1);
    int start = hufspec::probe_start(segTop, limit);                     // first run: the segment's tail, for its exit (zstd_hufspec.h)
    auto ld = [&](int off) -> u64 {
        if (off >= 0 && (u32)off + 8 <= sl) return ld64(sp + off);
        u64 v = 0;
        for (int i = 0; i < 8; ++i) { const int a = off + i; if (a >= 0 && (u32)a < sl) v |= (u64)sp[a] << (8 * i); }
        return v;
    };
    LT* tabp = (LT*)L->huf;
    auto tab = [&](u32 i) -> u32 { return tabp[i]; };
    u32 cnt = 0; int ex = start;
    bool dirty = true;
    for (int round = 0; ; ++round) {
        if (dirty) ex = hufspec::run_count(ld, tab, tl, start, limit, &cnt);
        const int prev = __shfl_up(ex, 1);
        const int want = sub == 0 ? segTop : prev;                       // (lane 0 of a stream starts at the stream's top)
        dirty = want != start;
        if (dirty) start = want;
        if (!any_lane(dirty)) break;
        if (round > 21) return false;                                    // (cannot happen: round r + 1 settles lane r of every stream)
    }
    DSTAGE(6);
    u32 inc = cnt;
    for (int d = 1; d < 16; d <<= 1) { const u32 t = (u32)__shfl_up((int)inc, d, 16); if (sub >= d) inc += t; }
    const u32 total = (u32)__shfl((int)inc, 15, 16);
    const int lastex = __shfl(ex, 15, 16);
    ok = ok && lastex == 0 && total == no;
    if (ballot(!ok) != 0) return false;
    const u32 off = oo + inc - cnt;
    hufspec::run_write(ld, tab, tl, start, limit,
                       [&](u32 i, u32 a, u32 b, u32 c, u32 d) { typedef u32x4 __attribute__((aligned(1))) U4; u32x4 v = {a, b, c, d}; *(__attribute__((address_space(1))) U4*)((gbyte*)out + off + i) = v; },
                       [&](u32 i, u32 v) { st32(out + off + i, v); }, [&](u32 i, u32 v) { out[off + i] = (u8)v; });
    DSTAGE(7);
    return true;
}

// ZSTD_buildSeqTable (zstd_decompress_block.c:659-705).  All lanes call it: lane 0 reads a transmitted description (the bit
// positions of its values depend on the values before them), the table itself is built by the wave, one symbol per lane.
__device__ __forceinline__ bool seq_table(u8* tab3, const int kind, u32* log, u32* have, u32 mode, const u8* src, u32* ip, u32 iend, u32 maxSym, u32 maxLog,
                          const s16* defNorm, u32 defMax, u32 defLog, DecLds* L, int lane)
{   // kind: 0 offsets, 1 match lengths, 2 literal lengths (how many additional bits a code reads)
    u8* occ = L->scratch; u16* next = (u16*)(L->scratch + 512);
    FseD* dt = (FseD*)tab3;                                              // built with four-byte cells, then packed in place
    typedef __attribute__((address_space(3))) u32 L32;
    typedef __attribute__((address_space(3))) u16 L16;
    typedef __attribute__((address_space(3))) u8 L8;
    auto cell3 = [&](u32 c) -> u32 {                                     // FseD (newState | symbol << 16 | nbBits << 24) -> the three-byte cell
        const u32 sym = (c >> 16) & 0xFFu;
        const u32 eb = kind == 0 ? sym : (kind == 1 ? (u32)kMLbits[sym < 53 ? sym : 0] : (u32)kLLbits[sym < 36 ? sym : 0]);
        // additional bits of the code (5) | code, low half (3) | state bits (4) | new state base (9) | code, high half (3): the
        // sequence loop takes both bit counts with ONE mask (cell & 0xF1F: the counts one byte apart, as its row sums want them)
        // and leaves the code where it is (cell & 0xE000E0, the sequence's bit position in the 13 bits between the halves)
        return (eb & 31u) | (sym & 7u) << 5 | ((c >> 24) & 15u) << 8 | (c & 0x1FFu) << 12 | ((sym >> 3) & 7u) << 21;
    };
    auto pack = [&](u32 tl) {                                            // tile by tile, upwards: the packed cells trail the cells still to read
        lbyte* t = (lbyte*)tab3;
        const u32 size = 1u << tl;
        for (u32 u0 = 0; u0 < size; u0 += kWave) {
            const u32 u = u0 + (u32)lane;
            u32 c = 0;
            if (u < size) c = *(L32*)(t + 4 * u);
            wave_fence();
            if (u < size) { const u32 v = cell3(c); *(L8*)(t + 3 * u) = (u8)v; *(L8*)(t + 3 * u + 1) = (u8)(v >> 8); *(L8*)(t + 3 * u + 2) = (u8)(v >> 16); }
            wave_fence();
        }
    };
    if (mode == 0) {
        const int norm = (u32)lane <= defMax ? (int)defNorm[lane] : 0;
        fse_build_dtable_wave(dt, norm, defMax, defLog, occ, next, lane);
        pack(defLog);
        if (lane == 0) { *log = defLog; *have = 1; }
        wave_fence();
        return true;
    }
    if (mode == 1) {
        if (*ip >= iend || src[*ip] > maxSym) return false;
        if (lane == 0) {
            const u32 v = cell3((u32)src[*ip] << 16);
            lbyte* t = (lbyte*)tab3;
            *(L8*)(t) = (u8)v; *(L8*)(t + 1) = (u8)(v >> 8); *(L8*)(t + 2) = (u8)(v >> 16);
            *log = 0; *have = 1;
        }
        wave_fence();
        (*ip)++; return true;
    }
    if (mode == 2) {
        u32 ms = maxSym, tl = 0, h = 0;
        // the description is parsed bit by bit on one lane: from a copy the wave stages in LDS (a table of <= 53 symbols and
        // table log <= 9 takes at most 62 bytes), not byte by byte from global memory
        const u32 avail = iend - *ip, stg = avail < 128u ? avail : 128u;
        for (u32 b = (u32)lane; b < stg; b += kWave) L->weights[b] = src[*ip + b];
        wave_fence();
        if (lane == 0) h = fse_read_ncount(L->norm, &ms, &tl, L->weights, stg, maxLog);
        wave_fence();
        h = uni(h); ms = uni(ms); tl = uni(tl);
        if (h == 0xFFFFFFFFu) return false;
        const int norm = (u32)lane <= ms ? (int)L->norm[lane] : 0;
        if (!fse_build_dtable_wave(dt, norm, ms, tl, occ, next, lane)) return false;
        pack(tl);
        if (lane == 0) { *log = tl; *have = 1; }
        wave_fence();
        *ip += h; return true;
    }
    return uni(*have) != 0;                                              // repeat: the caller fetches the table after the builds
}

// copy of `len` bytes inside dst from distance `offset` back (overlap allowed), whole wave
__device__ __forceinline__ void wave_match_copy(u8* d, u32 offset, u32 len, int lane)
{
    const u8* s = d - offset;
    if (offset >= kWave) {
        for (u32 r = 0; r < len; r += kWave) {
            const u32 i = r + (u32)lane;
            u8 v = 0;
            if (i < len) v = s[i];
            if (i < len) d[i] = v;
            wave_fence();
        }
    } else {
        u32 k = (u32)lane % offset; const u32 adv = kWave % offset;
        for (u32 i = (u32)lane; i < len; i += kWave) { d[i] = s[k]; k += adv; if (k >= offset) k -= offset; }
    }
    wave_fence();
}

// The offset history of 64 sequences at once (ZSTD_decodeSequence :1370-1400 runs it one sequence after the other).  What a
// sequence does to the history (r0, r1, r2) is a small map: every new entry is an old entry, an old entry minus one (offset
// value 3 after a zero literal length; the reference forces a 0 to 1, and max(max(x - a, 1) - b, 1) = max(x - a - b, 1) keeps
// that closed), or the sequence's own new offset.  Such maps compose, so a prefix scan over the lanes gives every sequence the
// history in front of it in six steps instead of 64 dependent ones on one lane; the sequence's offset is the front entry after
// its own map.  A map is three entries of {kind: old entry 0..2 or 3 = constant, k: how often one was subtracted, constant};
// meta holds (kind - j) & 3 and k per entry j, nine bits each, so that the identity is all zeros (what a row shift fills in).
struct RepMap { u32 meta, c0, c1, c2; };
__device__ __forceinline__ RepMap rep_compose(const RepMap E, const RepMap M)    // first E, then M
{
    RepMap R; u32 meta = 0;
    #pragma unroll
    for (int j = 0; j < 3; ++j) {
        const u32 fm = (M.meta >> (9 * j)) & 0x1FFu, km = ((fm & 3u) + (u32)j) & 3u;
        const u32 fe = (E.meta >> (9u * (km & 3u))) & 0x1FFu;                    // (km == 3: unused)
        const u32 ke = ((fe & 3u) + km) & 3u, kk = (fe >> 2) + (fm >> 2);
        const u32 ce = km == 0 ? E.c0 : (km == 1 ? E.c1 : E.c2);
        const u32 mc = j == 0 ? M.c0 : (j == 1 ? M.c1 : M.c2);
        const bool own = km == 3;
        const u32 f = own ? fm : (((ke - (u32)j) & 3u) | (kk << 2));
        const u32 c = own ? mc : ce;
        meta |= f << (9 * j);
        if (j == 0) R.c0 = c; else if (j == 1) R.c1 = c; else R.c2 = c;
    }
    R.meta = meta;
    return R;
}
__device__ __forceinline__ u32 rep_entry(const RepMap A, int j, u32 R0, u32 R1, u32 R2)
{
    const u32 f = (A.meta >> (9 * j)) & 0x1FFu, kind = ((f & 3u) + (u32)j) & 3u, k = f >> 2;
    const u32 c = j == 0 ? A.c0 : (j == 1 ? A.c1 : A.c2);
    const u32 base = kind == 3 ? c : (kind == 0 ? R0 : (kind == 1 ? R1 : R2));
    return k ? (base > k ? base - k : 1u) : base;
}
// val: the offset value of this lane's sequence (1..3 = repeat codes), ll its literal length; returns its offset and moves the
// history (wave-uniform R0..R2) past the tile.  Lanes without a sequence pass the identity.
__device__ __forceinline__ u32 rep_resolve(u32 val, u32 ll, bool active, u32& R0, u32& R1, u32& R2)
{
    const bool isRep = val <= 3;
    const u32 idx = val - 1 + (ll == 0 ? 1u : 0u);
    RepMap m; m.c0 = val - 3; m.c1 = 0; m.c2 = 0;
    m.meta = !isRep ? (3u | 3u << 9 | 3u << 18)
           : idx == 0 ? 0u
           : idx == 1 ? (1u | 3u << 9)
           : idx == 2 ? (2u | 3u << 9 | 3u << 18)
           :            (1u << 2 | 3u << 9 | 3u << 18);
    if (!active) { m.meta = 0; }
    auto step = [&](auto shifted) {
        RepMap E;
        E.meta = shifted(m.meta); E.c0 = shifted(m.c0); E.c1 = shifted(m.c1); E.c2 = shifted(m.c2);
        m = rep_compose(E, m);
    };
    step([](u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true); });
    step([](u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true); });
    step([](u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true); });
    step([](u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true); });
    step([](u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); });
    step([](u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); });
    const u32 off = rep_entry(m, 0, R0, R1, R2);
    const u32 n1 = rep_entry(m, 1, R0, R1, R2), n2 = rep_entry(m, 2, R0, R1, R2);
    R0 = lane_get(off, 63); R1 = lane_get(n1, 63); R2 = lane_get(n2, 63);
    return active ? off : 0u;
}

// one compressed block (ZSTD_decompressBlock_internal :2444-2552); returns decoded size or 0xFFFFFFFF
__device__ u32 decode_block(DecLds* L_, const u8* __restrict__ src_, u32 n_, u8* dst_, u32 opos_, u32 cap_, u32 blockMax_,
                            u8* litBuf_, int lane, const bool keep_)
{   // keep: another block of the frame follows and may reuse this block's tables (DecLds)
    // (a call, not inlined: inlined, the kernel's own state competes for the 80 registers and the tile loop spills ten times as
    //  often; the arguments of a call arrive in vector registers and are made scalar again here)
    DecLds* L = L_;                                      // (one call site: the compiler knows the LDS address as a constant - readfirstlane would hide it)
    const u8* src = uni_ptr(src_); const u32 n = uni(n_); u8* dst = uni_ptr(dst_);
    const u32 opos = uni(opos_), cap = uni(cap_), blockMax = uni(blockMax_); u8* litBuf = uni_ptr(litBuf_);
    const bool keep = uni(keep_ ? 1u : 0u) != 0;
    typedef __attribute__((address_space(3))) u32x4 LV4;
    auto park = [&](u32 at, const void* tab, u32 bytes) {                // LDS -> workspace
        gbyte* g = (gbyte*)litBuf + at;
        for (u32 i = (u32)lane; i < bytes / 16; i += kWave) *(__attribute__((address_space(1))) u32x4*)(g + 16 * i) = *(const LV4*)((const lbyte*)tab + 16 * i);
        wave_fence();
    };
    auto fetch = [&](void* tab, u32 at, u32 bytes) {                     // workspace -> LDS
        const gbyte* g = (const gbyte*)litBuf + at;
        wave_fence();
        for (u32 i = (u32)lane; i < bytes / 16; i += kWave) *(LV4*)((lbyte*)tab + 16 * i) = *(const __attribute__((address_space(1))) u32x4*)(g + 16 * i);
        wave_fence();
    };
#ifdef ZSTD_PROF
    unsigned long long dacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dprev = __builtin_amdgcn_s_memtime();
    unsigned long long dcnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // tiles, match levels, blocked-loop trips, whole-wave literal runs, whole-wave matches, sequences, one-per-lane matches
    struct Flush { unsigned long long* a; unsigned long long* c; int lane; __device__ ~Flush() { if (lane == 0) for (int k = 0; k < 8; ++k) { atomicAdd(&g_zstd_dprof[k], a[k]); atomicAdd(&g_zstd_dprof[8 + k], c[k]); } } } flush_{dacc, dcnt, lane};
#define DCOUNT(k, n) (dcnt[k] += (n))
#else
#define DCOUNT(k, n) ((void)0)
#endif
    // A failure is 0xFFFFFF00 | the reference's error code at the same check (zstd_errors.h:60-106), in the reference's order of checks.
    constexpr u32 kCorrupt = 0xFFFFFF00u | 20u, kDstSmall = 0xFFFFFF00u | 70u, kSrcSize = 0xFFFFFF00u | 72u, kLitHeader = 0xFFFFFF00u | 24u, kDictCorrupt = 0xFFFFFF00u | 30u;
    if (n < 2) return kCorrupt;                                          // MIN_CBLOCK_SIZE (zstd_decompress_block.c:174)
    const u32 room = cap - opos, writeMax = blockMax < room ? blockMax : room;   // expectedWriteSize (:169-170)
    u32 ip = 0, litSize = 0; const u8* litPtr = nullptr;
    {   // ---- literals (ZSTD_decodeLiteralsBlock :162-364) ----
        const u32 b0 = uni(src[0]), type = b0 & 3, fmt = (b0 >> 2) & 3;
        if (type < 2) {
            u32 lh;
            if (fmt == 0 || fmt == 2) { lh = 1; litSize = b0 >> 3; }
            else if (fmt == 1) { if (n < 2) return kCorrupt; lh = 2; litSize = (b0 | (uni(src[1]) << 8)) >> 4; }
            else { if (n < 3) return kCorrupt; lh = 3; litSize = (b0 | (uni(src[1]) << 8) | (uni(src[2]) << 16)) >> 4; }
            if (litSize > blockMax) return kCorrupt;
            if (litSize > writeMax) return kDstSmall;                    // (:290, :331)
            if (type == 0) { if (lh + litSize > n) return kCorrupt; litPtr = src + lh; ip = lh + litSize; }
            else {
                if (lh + 1 > n) return kCorrupt;
                const u8 v = (u8)uni(src[lh]);
                for (u32 i = (u32)lane; i < litSize; i += kWave) litBuf[i] = v;
                wave_fence();
                litPtr = litBuf; ip = lh + 1;
            }
        } else {
            u32 lh, csize; bool streams4 = true;
            if (fmt < 2) { if (n < 3) return kCorrupt; const u32 v = b0 | (uni(src[1]) << 8) | (uni(src[2]) << 16); lh = 3; streams4 = (fmt == 1); litSize = (v >> 4) & 0x3FF; csize = (v >> 14) & 0x3FF; }
            else if (fmt == 2) { if (n < 4) return kCorrupt; const u32 v = uni(ld32(src)); lh = 4; litSize = (v >> 4) & 0x3FFF; csize = v >> 18; }
            else { if (n < 5) return kCorrupt; const u32 v = uni(ld32(src)); lh = 5; litSize = (v >> 4) & 0x3FFFF; csize = (v >> 22) + (uni(src[4]) << 10); }
            if (type == 3 && !uni(L->haveHuf)) return kDictCorrupt;      // set_repeat without a table (:182), in front of every size check
            if (litSize > blockMax) return kCorrupt;                     // (:222)
            if (streams4 && litSize < 6) return kLitHeader;              // literals_headerWrong (:223-226)
            if (lh + csize > n) return kCorrupt;                         // (:227)
            if (litSize > writeMax) return kDstSmall;                    // (:228)
            if (litSize == 0) return kCorrupt;                           // (the Huffman decoder refuses an empty output, :267-274)
            u32 hoff = lh, hsz = csize;
            if (type == 2) {
                const u32 used = huf_read_table(L, src + hoff, hsz, (u16*)(litBuf + kBlockMax + 64), lane);
                if (used == 0xFFFFFFFFu) return kCorrupt;
                if (lane == 0) L->haveHuf = 1;
                hoff += used; hsz -= used;
            } else {
                if (!uni(L->haveHuf)) return kCorrupt;
                if (uni(L->hufLog) <= 11) fetch(L->huf, kDecHufPark, sizeof(L->huf));   // (the block that made it was not the frame's last)
            }
            wave_fence();
            bool ok = true;
            const u16* big = (const u16*)(litBuf + kBlockMax + 64);
            if (!streams4) ok = huf_decode_streams(L, litBuf, litSize, src + hoff, hsz, big, lane == 0, lane);
            else {
                if (litSize < 6 || hsz < 10) return kCorrupt;
                const u32 s1 = uni(ld16(src + hoff)), s2 = uni(ld16(src + hoff + 2)), s3 = uni(ld16(src + hoff + 4)), seg = (litSize + 3) / 4;
                if (6 + s1 + s2 + s3 > hsz || 3 * seg > litSize) return kCorrupt;
                const u32 s4 = hsz - 6 - s1 - s2 - s3;
                const u32 j = (u32)lane & 3;
                const u32 so = j == 0 ? 0 : (j == 1 ? s1 : (j == 2 ? s1 + s2 : s1 + s2 + s3));
                const u32 sl = j == 0 ? s1 : (j == 1 ? s2 : (j == 2 ? s3 : s4));
                const u32 no = j < 3 ? seg : litSize - 3 * seg;
                if (uni(L->hufLog) <= 11) {                                   // 16 lanes per stream
                    const u32 j4 = (u32)lane >> 4;
                    const u32 so4 = j4 == 0 ? 0 : (j4 == 1 ? s1 : (j4 == 2 ? s1 + s2 : s1 + s2 + s3));
                    const u32 sl4 = j4 == 0 ? s1 : (j4 == 1 ? s2 : (j4 == 2 ? s3 : s4));
                    if (!huf_decode_spec(L, litBuf, src + hoff + 6, so4, sl4, j4 * seg, j4 < 3 ? seg : litSize - 3 * seg, uni(L->hufLog), lane
#ifdef ZSTD_PROF
                                         , dacc, dprev
#endif
                                         )) {
                        // (cold) The speculative decoder wants every stream consumed exactly; the reference's fast loops - which run
                        // whenever the four streams have 16 / 8 / 8 / 8 bytes and the fourth segment is not empty, every decoding table
                        // being rescaled to log 11 for them (huf_decompress.c :184-270, :320-347, :1157) - make no such check.  Decode
                        // again, one lane per stream, and accept what leaves bits over.
                        if (!(s1 >= 16 && s2 >= 8 && s3 >= 8 && s4 >= 8 && 3 * seg < litSize)) return kCorrupt;
                        ok = huf_decode_streams(L, litBuf + j * seg, no, src + hoff + 6 + so, sl, big, lane < 4, lane, true);
                    }
                } else
                ok = huf_decode_streams(L, litBuf + j * seg, no, src + hoff + 6 + so, sl, big, lane < 4, lane,
                                        s1 >= 16 && s2 >= 8 && s3 >= 8 && s4 >= 8 && 3 * seg < litSize);
            }
            if (ballot(!ok) != 0) return kCorrupt;
            wave_fence();
            if (keep && type == 2 && uni(L->hufLog) <= 11) park(kDecHufPark, L->huf, sizeof(L->huf));
            litPtr = litBuf; ip = lh + csize;
        }
    }
    DSTAGE(0);
    litPtr = uni_ptr(litPtr); litSize = uni(litSize); ip = uni(ip);               // (scalar base: the tile's loads take a 32-bit offset each)
    // ---- sequences header (ZSTD_decodeSeqHeaders :707-790) ----
    if (ip >= n) return kSrcSize;                                        // MIN_SEQUENCES_SIZE: srcSize_wrong (:717)
    u32 nbSeq = uni(src[ip++]);
    if (nbSeq > 0x7F) {
        if (nbSeq == 0xFF) { if (ip + 2 > n) return kSrcSize; nbSeq = uni(ld16(src + ip)) + 0x7F00; ip += 2; }   // (:728)
        else { if (ip >= n) return kSrcSize; nbSeq = ((nbSeq - 0x80) << 8) + uni(src[ip++]); }                     // (:732)
    }
    nbSeq = uni(nbSeq);                                                          // (wave-uniform: loop counts stay in scalar registers)
    u32 out = opos, litPos = 0;
    if (nbSeq == 0) { if (ip != n) return kSrcSize; }                    // (:723)
    else {
        if (ip >= n) return kSrcSize;                                    // (:739)
        const u32 modes = uni(src[ip++]);                                // (the two reserved bits are not looked at, as in :741-743)
        const u32 mLL = modes >> 6, mOF = (modes >> 4) & 3, mML = (modes >> 2) & 3;
        if ((mLL == 3 || mOF == 3 || mML == 3) && !uni(L->haveLL | L->haveOF | L->haveML)) return kCorrupt;   // nothing to repeat
        if (!seq_table(L->ll3, 2, &L->llLog, &L->haveLL, mLL, src, &ip, n, 35, 9, kLLnorm, 35, 6, L, lane)) return kCorrupt;
        if (!seq_table(L->of3, 0, &L->ofLog, &L->haveOF, mOF, src, &ip, n, 31, 8, kOFnorm, 28, 5, L, lane)) return kCorrupt;
        if (!seq_table(L->ml3, 1, &L->mlLog, &L->haveML, mML, src, &ip, n, 52, 9, kMLnorm, 52, 6, L, lane)) return kCorrupt;
        // repeated tables, as the frame's last block with sequences left them (after the builds: building a table runs over the
        // places of the tables behind it)
        if (room == 0) return kDstSmall;                                 // (zstd_decompress_block.c:2087-2088)
        if (mLL == 3) fetch(L->ll3, kDecFsePark, sizeof(L->ll3));
        if (mOF == 3) fetch(L->of3, kDecFsePark + sizeof(L->ll3), sizeof(L->of3));
        if (mML == 3) fetch(L->ml3, kDecFsePark + sizeof(L->ll3) + sizeof(L->of3), sizeof(L->ml3));
        wave_fence();
        for (u32 i = (u32)lane; i < 36 + 53; i += kWave) {                       // length-code tables (they share LDS with the Huffman windows)
            if (i < 36) L->llTab[i] = kLLbase[i] | ((u32)kLLbits[i] << 24);
            else L->mlTab[i - 36] = kMLbase[i - 36] | ((u32)kMLbits[i - 36] << 24);
        }
        if (lane == 0) { L->zeroWord = 0; }
        wave_fence();
        // Sequences (ZSTD_decodeSequence :1339-1460), 64 per tile, then the wave executes the tile (:1007-1108).  The three FSE
        // states advance in lock step on three lanes - lane 0 offsets, 1 match lengths, 2 literal lengths (lanes 3-7 idle on a
        // cell of zeros so that every row shift reads a defined value) - instead of one after the other on one lane: one table
        // read and one state field per lane and sequence, the fields' bit positions from row-shift prefix sums (additional bits
        // are stored offset, match, literal; state bits literal, match, offset), branch-free.  What the chain does not need to
        // go on - the additional bits' values, the offset history - is done after the loop by all lanes (round 5).
        // Over-reading a damaged stream reads the zeros the window holds below bit 0, as BIT_reloadDStream leaves them.
        BitL b; b.pos = 0; b.wb = 0; b.c = 0; u32 sLL = 0, sOF = 0, sML = 0, okB = 1;
        const gbyte* bsp = (const gbyte*)(src + ip); const u32 bsn = n - ip;      // the sequence bitstream
        const lbyte* win = (const lbyte*)&L->bswin[0];
        int pos;                                                                 // unread bits of the sequence bitstream (wave-uniform)
        int wbase = 0;                                                           // stream byte offset of window byte 0
        auto refill = [&]() {                                                    // 48 lanes: the 768 bytes that end with the byte holding bit pos-1
            const int e = (pos + 7) >> 3, wb = e - 768;
            typedef __attribute__((address_space(3))) u32x4 LV;
            if (lane < 48) *(LV*)((lbyte*)&L->bswin[0] + 16 * lane) = stream16(bsp, bsn, wb + 16 * lane);
            wbase = wb;
            wave_fence();
        };
        {
            const int bits = (int)uni((u32)bit_stream_bits(src + ip, n - ip));
            if (bits < 0) return kCorrupt;
            pos = bits;
            refill();
        }
        if (lane == 0) {
            b.pos = pos; b.wb = wbase;
            b.grab(win); sLL = b.take(L->llLog); sOF = b.take(L->ofLog); sML = b.take(L->mlLog); if (b.pos < 0) okB = 0;
        }
        if (!uni(okB)) return kCorrupt;
        u32 R0 = uni(L->rep[0]), R1 = uni(L->rep[1]), R2 = uni(L->rep[2]);       // the offset history entering the next tile (wave-uniform)
        pos = (int)lane_get((u32)b.pos, 0);
        // per-lane roles of the sequence loop
        typedef __attribute__((address_space(3))) u32 L32;
        const u32 winA = (u32)(uintptr_t)win;
        u32 st = lane == 0 ? lane_get(sOF, 0) : lane == 1 ? lane_get(sML, 0) : lane == 2 ? lane_get(sLL, 0) : 0u;
        DSTAGE(1);
        lbyte* cb = (lbyte*)&L->seqBuild[0];                                      // history + output block of the tiles (lz_group.hip.h)
        bool hist_ok = false;
        for (u32 q0 = 0; q0 < nbSeq; q0 += kWave) {
            const u32 cnt = nbSeq - q0 < kWave ? nbSeq - q0 : kWave;
            if (q0) refill();
            // The loop carries only what the next state needs: the cell, the state bits and the bit positions.  The VALUES (base
            // of the code plus its additional bits) are read after the loop, every lane for its own sequence, from the code and
            // the position of the sequence's additional bits that the loop leaves behind (positions relative to the window: 13 bits).
            // Trip count and bit position are wave-uniform (readfirstlane): counters and branches stay in scalar registers.  Eight
            // lanes, not 64: an LDS instruction costs by its active lanes (all 64 lanes in this loop: 38 -> 70 ms for the bench).
            // (the lanes' roles are worked out anew for every tile: kept across the tile's execution they are three registers the
            //  executor needs - the opaque copy of the lane number keeps the compiler from hoisting them)
            int lane_o = lane; asm volatile("" : "+v"(lane_o));
            const u32 zeroA = (u32)(uintptr_t)(lbyte*)&L->zeroWord;
            const u32 tabA = lane_o == 0 ? (u32)(uintptr_t)(lbyte*)&L->of3[0] : lane_o == 1 ? (u32)(uintptr_t)(lbyte*)&L->ml3[0] : lane_o == 2 ? (u32)(uintptr_t)(lbyte*)&L->ll3[0] : zeroA;
            const u32 outA = lane_o == 0 ? (u32)(uintptr_t)(lbyte*)&L->seqOff[0] : lane_o == 1 ? (u32)(uintptr_t)(lbyte*)&L->seqML[0] : lane_o == 2 ? (u32)(uintptr_t)(lbyte*)&L->seqLL[0]
                                         : (u32)(uintptr_t)(lbyte*)&L->dumpWord;
            const u32 outStride = lane_o < 3 ? 4u : 0u;
            const int posU = (int)uni((u32)pos), wbU = (int)uni((u32)wbase);
            int posOut = posU;
            if (lane < 8) {
                const int wb8 = wbU * 8;
                int prel = posU - wb8;
                const u32 cntS = uni(cnt);
                u32 Slast = 0;
                for (u32 k = 0; k < cntS; ++k) {
                    const u32 cell = *(L32*)(uintptr_t)(tabA + __umul24(st, 3u));   // three-byte cells (seq_table); the top byte is the next cell's
                    // both bit counts in one register (additional bits <= 3 x 31 stay below 256): row-shift sums give lane j the
                    // counts of lanes j .. j+2, lane 0 the sequence's totals, and its high byte where lane j's state field begins
                    const u32 x = cell & 0xF1Fu;
                    const u32 P1 = x + (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x101, 0xF, 0xF, true);
                    const u32 P = P1 + (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x102, 0xF, 0xF, true);
                    const u32 tot = lane_get(P, 0);
                    const int Etot = (int)(tot & 0xFFu);
                    Slast = tot >> 8;
                    const int pe = (int)uni((u32)(prel - Etot));                 // (scalar: one vector subtraction is left)
                    const int q = pe - (int)((P >> 8) & 0xFFu);                  // the state field: <= 9 bits from window bit q upwards
                    const u32 w = lds_ld32((const lbyte*)(uintptr_t)(winA + (u32)(q >> 3)));
                    const u32 fs = __builtin_amdgcn_ubfe(w, (u32)q & 7u, x >> 8);
                    *(L32*)(uintptr_t)(outA + k * outStride) = (cell & 0xE000E0u) | ((u32)prel << 8);   // the code, and where the sequence's additional bits begin
                    st = __builtin_amdgcn_ubfe(cell, 12u, 9u) + fs;
                    prel = pe - (int)Slast;
                }
                if (q0 + cnt >= nbSeq) prel += (int)Slast;                       // the frame's last sequence reads no state bits
                posOut = prel + wb8;
            }
            pos = (int)uni((u32)posOut);
            wave_fence();
            // A tile that read below the stream's first bit: the block is refused here, before it is executed.  (The reference does
            // not notice - its reads wrap around the 64-bit container, bitstream.h:343-351 - and decodes garbage or fails later;
            // until round 6 this decoder read zeros there and could return other bytes than the reference's with a success value.)
            if (pos < 0) return kCorrupt;
            DSTAGE(2);
            // execute: prefix sums give every sequence its literal source and its output position
            u32 ll = 0, ml = 0, off = 0;
            if ((u32)lane < cnt) { ll = L->seqLL[lane]; ml = L->seqML[lane]; off = L->seqOff[lane]; }
            if ((u32)lane < cnt) {                                               // code and position -> value
                auto value = [&](u32 p, u32 base, u32 nb) -> u32 {               // base + the nb bits right below window bit p
                    const u32 q = p - nb;
                    const uint64_t x = lds_ld64((const lbyte*)(uintptr_t)(winA + (q >> 3)));
                    return base + __builtin_amdgcn_ubfe((u32)(x >> (q & 7)), 0u, nb);
                };
                // (additional bits are stored offset, match length, literal length from the top)
                auto code = [](u32 w) -> u32 { return ((w >> 5) & 7u) | ((w >> 18) & 0x38u); };   // (the record of the loop above)
                const u32 tl = L->llTab[code(ll)], tm = L->mlTab[code(ml)], so = code(off), top = (off >> 8) & 0x1FFFu;
                off = value(top, 1u << so, so);
                ml = value(top - so, tm & 0xFFFFFFu, tm >> 24);
                ll = value(top - so - (tm >> 24), tl & 0xFFFFFFu, tl >> 24);
            }
            off = rep_resolve(off, ll, (u32)lane < cnt, R0, R1, R2);
            if (q0 + cnt >= nbSeq && lane == 0) { L->rep[0] = R0; L->rep[1] = R1; L->rep[2] = R2; }
            const u32 incL = wave_scan_incl(ll), incO = wave_scan_incl(ll + ml);
            const u32 totL = lane_get(incL, 63), totO = lane_get(incO, 63);
            const u32 myLit = litPos + incL - ll, myOut = out + incO - (ll + ml);
            {   // A tile that cannot be executed fails as its FIRST failing sequence does in the reference (ZSTD_execSequenceEnd,
                // zstd_decompress_block.c:906-925): output room first (dstSize_tooSmall), then the literal buffer, then the offset.
                const bool in = (u32)lane < cnt;
                const bool fDst = in && (u64)out + incO > cap, fLit = in && incL > litSize - litPos, fOff = in && off > myOut + ll;
                const u64 fm = ballot(fDst || fLit || fOff);
                if (fm) return lane_get(fDst ? 1u : 0u, (int)__builtin_ctzll(fm)) ? kDstSmall : kCorrupt;
                if ((u64)(out - opos) + totO > blockMax) return kCorrupt;    // (a block that regenerates more than the frame's block size)
            }
            // The tile is executed by the LZ4 / Snappy decoders' group machinery (lz_group.hip.h): the sequences that fit an output
            // block in LDS go one per lane - literals and every match whose source ends in front of the block in ONE round of
            // loads, matches that read the block itself (or the 64 bytes of history in front of it) as LDS-to-LDS copies in
            // sequence order - and the block leaves with 16-byte stores: every output line is written once, and a chain of matches
            // that feed each other does not pay a trip to memory and back per link (4.4 dependency levels per tile on the bench
            // corpus; with 24 blocks per CU the half-written lines did not live in L2 until their last store, and the decoder
            // wrote 3x its output to HBM).  A sequence that does not fit a block, or whose match starts further in front of the
            // block than the history and still reaches into it, goes by the whole wave through global memory.  The block stands
            // where the tile's bitstream window and sequence arrays were (DecLds): both are dead by now.
            {
                const gbyte* gl = (const gbyte*)litPtr; gbyte* gd = (gbyte*)dst;
                const u32 relOut = incO - (ll + ml);                                      // tile-relative output offset
                u32 start = 0;
                while (start < cnt) {
                    const u32 g0 = lane_get(relOut, (int)start);                          // tile-relative output offset of this group
                    const bool in = (u32)lane >= start && (u32)lane < cnt;
                    const u32 lpos = relOut - g0, mst = lpos + ll, endrel = mst + ml;    // block-relative
                    const int32_t srel = (int32_t)mst - (int32_t)off;
                    const u64 fit = ballot(in && endrel <= kDecGrpBlock);                 // (a prefix of [start, cnt): endrel grows)
                    const u64 slow = ballot(in && ml != 0 && srel + (int32_t)ml > 0 && srel < -(int32_t)kGrpHist);
                    u32 ng = (u32)__builtin_popcountll(fit);
                    if (slow) { const u32 fsl = (u32)__builtin_ctzll(slow) - start; ng = fsl < ng ? fsl : ng; }
                    DCOUNT(1, 1);
                    if (ng == 0) {                                                        // one sequence by the whole wave
                        const u32 o_ = out + g0, l_ = lane_get(ll, (int)start), m_ = lane_get(ml, (int)start);
                        DCOUNT(4, 1);
                        wave_copy(gd + o_, gl + lane_get(myLit, (int)start), l_, lane);
                        wave_fence();
                        if (m_) wave_match_copy(dst + o_ + l_, lane_get(off, (int)start), m_, lane);
                        hist_ok = false;
                        start++;
                        continue;
                    }
                    const bool act = in && (u32)lane < start + ng;
                    const u32 total = lane_get(endrel, (int)(start + ng - 1));
                    (void)lz_group_execute<true, true>(gl, gd, cb, hist_ok, out + g0, ng, act, ll, ml, off, myLit, lpos, mst, total);
                    start += ng;
                }
                DCOUNT(0, 1); DCOUNT(5, cnt);
            }
            litPos += totL; out += totO;
            DSTAGE(4);
        }
        if (pos > 0) return kCorrupt;                                         // bitstream not fully consumed
        if (keep) park(kDecFsePark, L->ll3, 3840);
    }
    const u32 rest = litSize - litPos;
    if ((u64)out + rest > cap) return kDstSmall;                         // last literals (:1588, :1763)
    if ((u64)(out - opos) + rest > blockMax) return kCorrupt;
    wave_copy(dst + out, litPtr + litPos, rest, lane);
    wave_fence();
    out += rest;
    return out - opos;
}

__device__ __forceinline__ int zstd_decompress_wave(const u8* __restrict__ src_, const u32 n_, u8* dst_, const u32 cap_, DecLds* L,
                                    u8* litBuf_, int lane)
{
    // Everything that steers the decoder is wave-uniform, and said so (readfirstlane): what is read from the stream - frame and
    // block headers, section sizes, modes - would otherwise count as divergent, and with it every position, every loop and every
    // address derived from it (vector registers, exec-mask loops, 64-bit vector address arithmetic; the spills of that cost the
    // decoder more HBM traffic than its output).
    const u8* src = uni_ptr(src_); const u32 n = uni(n_); u8* dst = uni_ptr(dst_); const u32 cap = uni(cap_); u8* litBuf = uni_ptr(litBuf_);
    // Failures are -(the reference's error code) at the same check, in the reference's order (zstd_errors.h:60-106;
    // ZSTD_decompressMultiFrame zstd_decompress.c:1114-1197, ZSTD_decompressFrame :1003-1111, ZSTD_getFrameHeader_advanced :440-560).
    constexpr int kPrefix = -10, kUnsupported = -14, kWindow = -16, kCorrupt = -20, kChecksum = -22, kDictWrong = -32, kDstSmall = -70, kSrcSize = -72;
    u32 ip = 0, op = 0;
    bool more = false;                                                           // a frame has been decoded (moreThan1Frame)
    while (n - ip >= 5) {                                                        // ZSTD_startingInputLength
        const u32 magic = uni(ld32(src + ip));
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {                              // skippable frame (readSkippableFrameSize :571-586)
            if (n - ip < 8) return kSrcSize;
            const u32 sz = uni(ld32(src + ip + 4));
            if ((u32)(sz + 8) < sz) return kUnsupported;
            if ((u64)sz + 8 > n - ip) return kSrcSize;
            ip += 8 + sz; continue;
        }
        if (n - ip < 9) return kSrcSize;                                         // frame header minimum + one block header (:1017-1019)
        const u32 fhd = uni(src[ip + 4]);
        const u32 fcsId = fhd >> 6, single = (fhd >> 5) & 1, hasCk = (fhd >> 2) & 1, dictIdSz = fhd & 3;
        const u32 dsz = dictIdSz == 3 ? 4 : dictIdSz, fl = fcsId == 0 ? single : (fcsId == 1 ? 2u : (fcsId == 2 ? 4u : 8u));
        const u32 fhs = 5 + (single ? 0u : 1u) + dsz + fl;                       // ZSTD_frameHeaderSize_internal
        if (n - ip < fhs + 3) return kSrcSize;                                   // (:1025-1026)
        if (magic != 0xFD2FB528u) return more ? kSrcSize : kPrefix;              // prefix_unknown; behind a decoded frame: srcSize_wrong (:1174-1182)
        if (fhd & 0x08) return kUnsupported;                                     // reserved bit (:500-502)
        ip += 5;
        u64 windowSize = 0, fcs = 0; bool haveFcs = false;
        if (!single) {
            const u32 wb = uni(src[ip++]), wl = (wb >> 3) + 10;
            if (wl > 31) return kWindow;                                         // ZSTD_WINDOWLOG_MAX (:516-521)
            windowSize = 1ull << wl; windowSize += (windowSize >> 3) * (wb & 7);
        }
        {   u32 id = 0;
            for (u32 k = 0; k < dsz; ++k) id |= (u32)src[ip + k] << (8 * k);
            id = uni(id);
            ip += dsz;
            if (fcsId == 0) { if (single) { fcs = uni(src[ip++]); haveFcs = true; } }
            else if (fcsId == 1) { fcs = (u64)uni(ld16(src + ip)) + 256; ip += 2; haveFcs = true; }
            else if (fcsId == 2) { fcs = uni(ld32(src + ip)); ip += 4; haveFcs = true; }
            else { const u64 f8 = ld64(src + ip); fcs = ((u64)uni((u32)(f8 >> 32)) << 32) | uni((u32)f8); ip += 8; haveFcs = true; }
            if (id != 0) return kDictWrong;                                      // a dictionary ID and no dictionary (ZSTD_decodeFrameHeader :964-965)
        }
        if (single) windowSize = fcs;
        const u32 frameStart = op;
        if (lane == 0) { L->rep[0] = 1; L->rep[1] = 4; L->rep[2] = 8; L->haveLL = L->haveOF = L->haveML = L->haveHuf = 0; }
        wave_fence();
        const u32 blockMax = windowSize < kBlockMax ? (u32)windowSize : kBlockMax;
        for (;;) {                                                               // ZSTD_decompressFrame :1031-1090
            if (n - ip < 3) return kSrcSize;                                     // ZSTD_getcBlockSize (:871-872)
            const u32 bh = uni(src[ip] | ((u32)src[ip + 1] << 8) | ((u32)src[ip + 2] << 16)); ip += 3;
            const u32 last = bh & 1, type = (bh >> 1) & 3, bsz = bh >> 3;
            if (type == 3) return kCorrupt;                                      // bt_reserved (:880)
            if ((type == 1 ? 1u : bsz) > n - ip) return kSrcSize;                // (:1040)
            if (type == 1) {
                if (bsz > cap - op) return kDstSmall;                            // ZSTD_setRleBlock (:911)
                const u8 v = (u8)uni(src[ip]);
                for (u32 i = (u32)lane; i < bsz; i += kWave) dst[op + i] = v;
                wave_fence();
                ip += 1; op += bsz;
            } else if (type == 0) {
                if (bsz > cap - op) return kDstSmall;                            // ZSTD_copyRawBlock (:895)
                wave_copy(dst + op, src + ip, bsz, lane);
                wave_fence();
                ip += bsz; op += bsz;
            } else {
                if (bsz > blockMax) return kSrcSize;                             // ZSTD_decompressBlock_internal (zstd_decompress_block.c:2064)
                const u32 r = decode_block(L, src + ip, bsz, dst + frameStart, op - frameStart, cap - frameStart, blockMax, litBuf, lane, last == 0);
                if (r >= 0xFFFFFF00u) return -(int)(r & 0xFFu);
                ip += bsz; op += r;
            }
            if (last) break;
        }
        if (haveFcs && (u64)(op - frameStart) != fcs) return kCorrupt;           // (:1092-1095)
        if (hasCk) { if (n - ip < 4) return kChecksum; ip += 4; }                // the value itself is checked by the host path (rap.hip, checksums_ok)
        more = true;
    }
    if (n - ip) return kSrcSize;                                                 // input not entirely consumed (:1194)
    return (int)op;
}

}  // namespace

// Double-fast (DF): a workgroup is kDfWaves INDEPENDENT wavefronts (own chunk, own tables and workspace, own 4 KB of slot-reuse
// scratch; no barrier after the first instruction).  The finder needs almost no LDS and waits on HBM most of the time, the
// entropy stage needs 17.5 KB for a few percent of a chunk's time: the waves of a workgroup share ONE entropy area behind a
// lock word, so that LDS no longer bounds the resident chunks (9 -> 12 per CU, the register bound).
template <bool DF>
__global__ void __launch_bounds__(DF ? 64 * DF_WAVES : 64) __attribute__((amdgpu_waves_per_eu(DF ? 3 : 1, DF ? 4 : 8))) zstd_compress_kernel(aoclgpu_batch b, uint8_t* wsBase, uint32_t ldsBytes,
                                                           uint32_t first, uint32_t count, uint8_t* tabBase, uint32_t tabBytes, uint32_t* counter)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t wv = DF ? threadIdx.x >> 6 : 0u, unit = DF ? blockIdx.x * kDfWaves + wv : blockIdx.x;
    Workspace* ws = reinterpret_cast<Workspace*>(wsBase) + unit;
    u32* tabs = tabBase ? reinterpret_cast<u32*>(tabBase + (size_t)unit * tabBytes) : nullptr;          // double-fast: this wave's two tables
    const int lane = lane_id();
    uint8_t* mine = lds; void* entLds = nullptr; u32* lock = nullptr;
    if constexpr (DF) {
        const uint32_t ent = (uint32_t)((sizeof(EntropyLds) + 255) & ~(size_t)255);
        entLds = lds; mine = lds + ent + wv * kDfScratchBytes; lock = reinterpret_cast<u32*>(lds + ent + kDfWaves * kDfScratchBytes);
        if (threadIdx.x == 0) *lock = 0;
        __syncthreads();
        ldsBytes = kDfScratchBytes;
    }
    // chunks are handed out through a counter (zeroed by the launcher): a chunk's duration varies by a factor of several with
    // its content, and a static stride leaves the workgroups that drew the short ones idle at the end of the launch
    for (;;) {
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(counter, 1u);
        c = uni(c);
        if (c >= count) break;
        c += first;
        const ChunkView v = compress_view(b, c);
        const int r = zstd_compress_wave<DF>((const uint8_t*)v.src, v.src_len, (uint8_t*)v.dst, v.dst_cap, mine, ldsBytes, ws, tabs, lane, b.reserved, entLds, lock);
        if (lane == 0) b.out_len[c] = r;
        wave_fence();
    }
}

// Double-fast as two kernels (large batches).  The finder needs 36 VGPRs and 4 KB of LDS - the hardware's 32 wavefronts per CU
// fit - and spends its life waiting for tables in HBM; fused with the entropy stage (135 VGPRs) only 12 chunks per CU are in
// flight.  So: zstd_df_find_kernel parses a slab of chunks into per-chunk workspaces, zstd_df_entropy_kernel turns every
// workspace into its frame.
__global__ void __launch_bounds__(64) zstd_df_find_kernel(aoclgpu_batch b, uint8_t* wsBase, uint32_t first, uint32_t count, uint8_t* tabBase, uint32_t tabBytes,
                                                          uint32_t* counter, uint32_t* meta)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[kDfScratchBytes];
    u32* tabs = reinterpret_cast<u32*>(tabBase + (size_t)blockIdx.x * tabBytes);
    const int lane = lane_id();
    for (;;) {
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(counter, 1u);
        c = uni(c);
        if (c >= count) break;
        const ChunkView v = compress_view(b, first + c);
        (void)zstd_compress_wave<true, 1>((const uint8_t*)v.src, v.src_len, (uint8_t*)v.dst, v.dst_cap, lds, kDfScratchBytes, reinterpret_cast<Workspace*>(wsBase) + c,
                                          tabs, lane, b.reserved, nullptr, nullptr, meta + 2 * (size_t)c);
        wave_fence();
    }
}

__global__ void __launch_bounds__(64) zstd_df_entropy_kernel(aoclgpu_batch b, uint8_t* wsBase, uint32_t ldsBytes, uint32_t first, uint32_t count,
                                                             uint32_t* counter, uint32_t* meta)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = lane_id();
    for (;;) {
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(counter, 1u);
        c = uni(c);
        if (c >= count) break;
        const ChunkView v = compress_view(b, first + c);
        const int r = zstd_compress_wave<true, 2>((const uint8_t*)v.src, v.src_len, (uint8_t*)v.dst, v.dst_cap, lds, ldsBytes, reinterpret_cast<Workspace*>(wsBase) + c,
                                                  nullptr, lane, b.reserved, nullptr, nullptr, meta + 2 * (size_t)c);
        if (lane == 0) b.out_len[first + c] = r;
        wave_fence();
    }
}

#ifndef ZSTD_DEC_WAVES
#define ZSTD_DEC_WAVES 6
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ZSTD_DEC_WAVES, ZSTD_DEC_WAVES))) zstd_decompress_kernel(aoclgpu_batch b, uint8_t* litBase, uint32_t* counter)
{
    __shared__ DecLds L;
    uint8_t* litBuf = litBase + (size_t)blockIdx.x * kDecWs;       // literals, then room for a 12-bit Huffman table
    const int lane = lane_id();
    for (;;) {                                                     // dynamic hand-out, as in zstd_compress_kernel
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(counter, 1u);
        c = uni(c);
        if (c >= b.nchunks) break;
        const ChunkView v = decompress_view(b, c);
        const int r = zstd_decompress_wave((const uint8_t*)v.src, v.src_len, (uint8_t*)v.dst, v.dst_cap, &L, litBuf, lane);
        if (lane == 0) b.out_len[c] = r;
        wave_fence();
    }
}

// (9 frames per CU by LDS need a SIMD to hold three waves: at most 170 VGPRs)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) zstd_frame_kernel(aoclgpu_zstd_frames f, uint8_t* wsBase, uint8_t* tabBase, uint32_t tabBytes, const uint32_t* sched)
{
    __shared__ FrameLds L;
    Workspace* ws = reinterpret_cast<Workspace*>(wsBase) + blockIdx.x;
    u32* tab = reinterpret_cast<u32*>(tabBase + (size_t)blockIdx.x * tabBytes);
    const int lane = lane_id();
    for (uint32_t i = blockIdx.x; i < f.nframes; i += gridDim.x) {
        const uint64_t so = f.src_off[i], sn = f.src_off[i + 1] - so, d0 = f.dst_off[i], dc = f.dst_off[i + 1] - d0;
        const long long r = zstd_frame_wave<false>(f.src + so, sn, f.dst + d0, dc, f.strategy, f.windowLog, f.hashLog, f.chainLog, f.minMatch, f.targetLength, &L, ws, tab, lane, sched);
        if (lane == 0) f.out_len[i] = r;
        wave_fence();
    }
}

// the same with the fast finder's table and source ring in LDS (dynamic: 4 << hashLog bytes + ring + FrameLds): one frame per CU
__global__ void __launch_bounds__(64) zstd_frame_lds_kernel(aoclgpu_zstd_frames f, uint8_t* wsBase, const uint32_t* sched)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    Workspace* ws = reinterpret_cast<Workspace*>(wsBase) + blockIdx.x;
    FrameLds* L = reinterpret_cast<FrameLds*>(lds + ((size_t)4 << f.hashLog) + kSrcRingBytes);
    const int lane = lane_id();
    for (uint32_t i = blockIdx.x; i < f.nframes; i += gridDim.x) {
        const uint64_t so = f.src_off[i], sn = f.src_off[i + 1] - so, d0 = f.dst_off[i], dc = f.dst_off[i + 1] - d0;
        const long long r = zstd_frame_wave<true>(f.src + so, sn, f.dst + d0, dc, f.strategy, f.windowLog, f.hashLog, f.chainLog, f.minMatch, f.targetLength, L, ws,
                                                  reinterpret_cast<u32*>(lds), lane, sched);
        if (lane == 0) f.out_len[i] = r;
        wave_fence();
    }
}

// pair schedule of the fast parser for a step size other than 2 (FastSched's construction at run time): out[0..L) offsets,
// out[L..2L) gaps to the repcode probe, out[2L..3L) the value of `step`, L = kSchedLen
__global__ void zstd_fast_sched_kernel(uint32_t* out, uint32_t stepSize)
{
    if (blockIdx.x || threadIdx.x) return;
    u32 step = stepSize, nextStep = 128, A = 0, sig = stepSize;
    for (u32 m = 0; m < kSchedLen; ++m) {
        out[m] = A; out[kSchedLen + m] = sig; out[2 * kSchedLen + m] = step;
        const u32 A2 = A + sig; sig = step;
        if (A2 + step >= nextStep) { step++; nextStep += 128; }
        A = A2;
    }
}
constexpr size_t kFastSchedBytes = 3 * kSchedLen * sizeof(uint32_t);

size_t zstd_workspace_bytes() { return sizeof(Workspace); }
size_t zstd_frame_lds_bytes() { return sizeof(FrameLds); }
uint32_t zstd_decoder_lds_bytes() { return (uint32_t)sizeof(DecLds); }
size_t zstd_entropy_lds_bytes() { return sizeof(EntropyLds); }

}  // namespace aoclgpu

// ------------------------------------------------------------------------------------------------
// C-ABI launchers
// ------------------------------------------------------------------------------------------------
namespace {
using aoclgpu::get_workspace;
uint32_t lds_for_chunk(uint64_t n, int level = 1)
{   // hash table bytes of the level's parameters for an n-byte chunk, never less than the entropy-stage LDS
    uint32_t need;
    if (level >= 3) {                                    // double-fast: tables in HBM; LDS = one entropy area + the waves' slot-reuse scratch + lock word
        const uint32_t ent = ((uint32_t)aoclgpu::zstd_entropy_lds_bytes() + 255u) & ~255u;
        return ent + aoclgpu::kDfWaves * aoclgpu::kDfScratchBytes + 256u;
    }
    else if (n <= 16384) {
        uint32_t srcLog = 6; if (n >= 64) { srcLog = 0; uint64_t t = n - 1; while (t) { srcLog++; t >>= 1; } }
        const uint32_t w = srcLog < 14 ? srcLog : 14, hl = (15 < w + 1) ? 15 : w + 1;
        need = (2u << hl) + aoclgpu::kSrcRingBytes;
    } else {
        const uint32_t hl = level == 2 ? 15 : 13;
        need = (2u << hl) + (1u << hl) / 8 + aoclgpu::kSrcRingBytes;
    }
    const uint32_t ent = (uint32_t)aoclgpu::zstd_entropy_lds_bytes();
    need = need > ent ? need : ent;
    return (need + 255u) & ~255u;
}
}  // namespace

// The match finder's sequences for ONE chunk already in device memory (used by the sequence producer below and by the
// debugging tools):
// out_host[0] = nbSeq, out_host[1] = litSize, then {litLength, offBase, matchLength} per sequence.
namespace aoclgpu {
__global__ void __launch_bounds__(64) zstd_block_sequences_kernel(const uint8_t* src, uint32_t n, uint8_t* wsBase, uint32_t* out, uint32_t maxSeq, uint32_t expf)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    Workspace* ws = reinterpret_cast<Workspace*>(wsBase);
    const int lane = lane_id();
    u32 hlog, mls, litSize = 0, nbSeq;
    level1_params(n, &hlog, &mls);
    if (mls == 5) nbSeq = fast_block<5, u16, false>((const gbyte*)src, n, hlog, (lbyte*)lds, ws, &litSize, lane, expf);
    else nbSeq = fast_block<6, u16, true>((const gbyte*)src, n, hlog, (lbyte*)lds, ws, &litSize, lane, expf);
    wave_fence();
    if (lane == 0) { out[0] = nbSeq; out[1] = litSize; }
    for (u32 i = (u32)lane; i < nbSeq && i < maxSeq; i += kWave) { out[2 + 3 * i] = ws->seqLL[i]; out[3 + 3 * i] = ws->seqOff[i]; out[4 + 3 * i] = ws->seqML[i]; }
}
}  // namespace aoclgpu
extern "C" int aoclgpu_zstd_block_sequences(const void* src_dev, uint32_t n, uint32_t* out_host, uint32_t maxSeq, uint32_t expf)
{
    if (n < 7 || n > (128u << 10)) return AOCLGPU_ERR_ARG;
    aoclgpu::WsGuard guard;                               // the scratch is used until the synchronous copy below returns
    void* ws = get_workspace(guard, aoclgpu::zstd_workspace_bytes(), nullptr);
    uint32_t* out = nullptr;
    const size_t bytes = (2 + 3 * (size_t)maxSeq) * sizeof(uint32_t);
    if (!ws || hipMalloc((void**)&out, bytes) != hipSuccess) return AOCLGPU_ERR_HIP;
    const uint32_t lds = lds_for_chunk(n);
    if (lds > 65536u) (void)hipFuncSetAttribute((const void*)aoclgpu::zstd_block_sequences_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(aoclgpu::zstd_block_sequences_kernel, dim3(1), dim3(64), lds, nullptr, (const uint8_t*)src_dev, n, (uint8_t*)ws, out, maxSeq, expf);
    const hipError_t e = hipMemcpy(out_host, out, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(out);
    return e == hipSuccess ? AOCLGPU_OK : AOCLGPU_ERR_HIP;
}

// Block-level external sequence producer for libzstd (ZSTD_sequenceProducer_F, /root/reference algos/zstd/lib/zstd.h:4065-4098;
// consumed by ZSTD_registerSequenceProducer, zstd_compress.c:3346-3406, 7370-7386): the GPU match finder parses the block,
// libzstd on the host does the entropy stage.  offBase values are resolved to real offsets with the format's repcode
// rules (history {1, 4, 8} per block, as each block is parsed without history); the parse ends with a block delimiter
// that carries the last literals.  Any failure returns ZSTD_SEQUENCE_PRODUCER_ERROR so that zstd can fall back.
extern "C" size_t aoclgpu_zstd_sequence_producer(void* state, void* outSeqs, size_t outSeqsCapacity, const void* src, size_t srcSize,
                                                 const void* dict, size_t dictSize, int compressionLevel, size_t windowSize)
{
    (void)state; (void)dict; (void)compressionLevel; (void)windowSize;
    struct Seq { unsigned offset, litLength, matchLength, rep; };
    const size_t kErr = (size_t)-1;
    Seq* o = (Seq*)outSeqs;
    if (dictSize != 0 || srcSize > (128u << 10) || !o || outSeqsCapacity < 1) return kErr;
    if (srcSize < 7) { o[0] = Seq{0, (unsigned)srcSize, 0, 0}; return 1; }          // below the match finder's minimum
    static std::mutex mu;
    static void* dsrc = nullptr; static uint32_t* hout = nullptr;
    std::lock_guard<std::mutex> lk(mu);
    const uint32_t maxSeq = (128u << 10) / 3 + 16;
    if (!dsrc && hipMalloc(&dsrc, (128u << 10) + 64) != hipSuccess) { dsrc = nullptr; return kErr; }
    if (!hout) hout = (uint32_t*)malloc((2 + 3 * (size_t)maxSeq) * sizeof(uint32_t));
    if (!hout || hipMemcpy(dsrc, src, srcSize, hipMemcpyHostToDevice) != hipSuccess) return kErr;
    if (aoclgpu_zstd_block_sequences(dsrc, (uint32_t)srcSize, hout, maxSeq, 0) != AOCLGPU_OK) return kErr;
    const uint32_t nb = hout[0];
    if ((size_t)nb + 1 > outSeqsCapacity || nb > maxSeq) return kErr;
    uint32_t rep1 = 1, rep2 = 4, rep3 = 8;
    size_t consumed = 0;
    for (uint32_t i = 0; i < nb; ++i) {
        const uint32_t ll = hout[2 + 3 * i], ob = hout[3 + 3 * i], ml = hout[4 + 3 * i];
        uint32_t off;
        if (ob > 3) { off = ob - 3; rep3 = rep2; rep2 = rep1; rep1 = off; }
        else if (ob == 1 && ll > 0) off = rep1;                                       // the fast strategy only emits repcode 1
        else if (ob == 1) { off = rep2; rep2 = rep1; rep1 = off; }                    // litLength == 0 shifts the repcodes by one
        else return kErr;
        o[i] = Seq{off, ll, ml, 0};
        consumed += (size_t)ll + ml;
    }
    (void)rep3;
    if (consumed > srcSize) return kErr;
    o[nb] = Seq{0, (unsigned)(srcSize - consumed), 0, 0};
    return (size_t)nb + 1;
}

#ifdef ZSTD_TRACE
extern "C" int aoclgpu_debug_zstd_trace(unsigned long long* host, unsigned int* count)
{
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(aoclgpu::g_ztrace), 8 * 4096 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(count, HIP_SYMBOL(aoclgpu::g_ztrace_n), sizeof(unsigned int)) != hipSuccess) return -1;
    unsigned int z = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(aoclgpu::g_ztrace_n), &z, sizeof(z));
    return 0;
}
#endif
#ifdef ZSTD_PROF
extern "C" int aoclgpu_debug_zstd_dprof(unsigned long long* host16)
{
    if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(aoclgpu::g_zstd_dprof), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(aoclgpu::g_zstd_dprof), z, sizeof(z));
    return 0;
}
extern "C" int aoclgpu_debug_zstd_eprof(unsigned long long* host16)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(aoclgpu::g_zstd_eprof), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(aoclgpu::g_zstd_eprof), z, sizeof(z));
    return 0;
}
extern "C" int aoclgpu_debug_zstd_gprof(unsigned long long* host4)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(host4, HIP_SYMBOL(aoclgpu::g_zstd_gprof), 4 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[4] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(aoclgpu::g_zstd_gprof), z, sizeof(z));
    return 0;
}
extern "C" int aoclgpu_debug_zstd_hprof(unsigned long long* host8)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(aoclgpu::zs::g_zstd_hprof), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(aoclgpu::zs::g_zstd_hprof), z, sizeof(z));
    return 0;
}
extern "C" int aoclgpu_debug_zstd_prof(unsigned long long* host16)
{
    if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(aoclgpu::g_zstd_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(aoclgpu::g_zstd_prof), z, sizeof(z));
    return 0;
}
#endif
namespace {
// Double-fast on a large batch: finder kernel + entropy kernel per slab of chunks (see zstd_df_find_kernel).  A chunk's
// sequences and literals wait in a workspace of its own between the two kernels (1 MB per chunk), so the batch goes through in
// slabs of at most AOCLGPU_ZSTD_DF_SLAB_MB (default: 32 GiB or a quarter of the device's memory, whichever is smaller).
// Returns AOCLGPU_OK when enqueued, 1 when the caller should use the fused kernel (batch too small to gain, scratch not
// available), an error code otherwise.
std::atomic<int> g_df_last_slabs{0};                      // slabs of the newest double-fast launch, 0: the fused kernel ran
int df_split_launch(const aoclgpu_batch* b, uint32_t first, uint32_t count, size_t tabBytes, const aoclgpu::WsGuard& guard, hipStream_t s)
{
    g_df_last_slabs.store(0);
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 1;
    const uint32_t cus = (uint32_t)prop.multiProcessorCount;
    if (const char* e = getenv("AOCLGPU_ZSTD_DF_SPLIT")) { if (!strcmp(e, "0")) return 1; }
    if (count <= 16u * cus) return 1;                     // every chunk gets its own wavefront at once anyway
    if (b->src_off || b->chunk_bytes < (64u << 10)) return 1;   // a workspace is 1 MB whatever the chunk's size: not for batches of small chunks
    int perA = 0, perB = 0;
    const uint32_t ldsB = ((uint32_t)aoclgpu::zstd_entropy_lds_bytes() + 255u) & ~255u;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perA, (const void*)aoclgpu::zstd_df_find_kernel, 64, 0) != hipSuccess || perA < 1 ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&perB, (const void*)aoclgpu::zstd_df_entropy_kernel, 64, ldsB) != hipSuccess || perB < 1) return 1;
    {
        const int byLds = (int)((160u << 10) / (((ldsB + 1279u) / 1280u) * 1280u));
        if (byLds >= 1 && byLds < perB) perB = byLds;
        if (perA > 32) perA = 32;
    }
    size_t budget = (size_t)32 << 30;
    if (prop.totalGlobalMem / 4 < budget) budget = prop.totalGlobalMem / 4;
    {   // what the device can actually give (another allocator, e.g. torch's caching allocator, may hold most of it): the scratch
        // this stream already owns plus the free memory, less a margin
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
            const size_t margin = (size_t)512 << 20, have = aoclgpu::workspace_held(guard, s);
            const size_t avail = have + (freeB > margin ? freeB - margin : 0);
            if (avail < budget) budget = avail;
        } else (void)hipGetLastError();
    }
    if (const char* e = getenv("AOCLGPU_ZSTD_DF_SLAB_MB")) { const long long mb = atoll(e); if (mb > 0) budget = (size_t)mb << 20; }
    const size_t wsz = aoclgpu::zstd_workspace_bytes();
    // the allocation is more than the per-chunk workspaces: the resident workgroups' hash tables (perA x CUs tables of
    // 384-768 KiB), 8 bytes of meta per chunk, the slab counters and the alignment of the three regions - all of it has to
    // fit the budget, or the hipMalloc below fails exactly when free memory is what limits the slab
    const size_t fixed = (size_t)perA * cus * tabBytes + (size_t)64 * 256 + 3 * 256;
    if (budget <= fixed) return 1;
    const size_t fit = (budget - fixed) / (wsz + 8);
    uint32_t slab = (uint32_t)(fit < count ? fit : count);
    if (slab < 8u * cus) return 1;                       // too few chunks in flight to beat the fused kernel
    // equal slabs (the last one is not a remainder of a few chunks)
    const uint32_t nslab = (count + slab - 1) / slab;
    slab = (count + nslab - 1) / nslab;
    uint32_t gridA = (uint32_t)perA * cus; if (gridA > slab) gridA = slab;
    uint32_t gridB = (uint32_t)perB * cus; if (gridB > slab) gridB = slab;
    const size_t seqBytes = ((size_t)slab * wsz + 255) & ~(size_t)255;
    const size_t tabOff = seqBytes, metaOff = (tabOff + (size_t)gridA * tabBytes + 255) & ~(size_t)255;
    const size_t ctrOff = (metaOff + (size_t)slab * 8 + 255) & ~(size_t)255;
    if (nslab > 64) return 1;                            // (the counters' share of `fixed` above)
    uint8_t* ws = (uint8_t*)get_workspace(guard, ctrOff + (size_t)nslab * 256, s);
    if (!ws) return 1;
    g_df_last_slabs.store((int)nslab);
    if (hipMemsetAsync(ws + ctrOff, 0, (size_t)nslab * 256, s) != hipSuccess) { aoclgpu::set_error_msg("zstd: hipMemsetAsync failed"); return AOCLGPU_ERR_HIP; }
    for (uint32_t k = 0; k < nslab; ++k) {
        const uint32_t f0 = first + k * slab, m = (k + 1) * slab <= count ? slab : count - k * slab;
        uint32_t* ctr = (uint32_t*)(ws + ctrOff + (size_t)k * 256);
        hipLaunchKernelGGL(aoclgpu::zstd_df_find_kernel, dim3(gridA < m ? gridA : m), dim3(64), 0, s, *b, ws, f0, m, ws + tabOff, (uint32_t)tabBytes, ctr, (uint32_t*)(ws + metaOff));
        if (int rc = aoclgpu::check_launch("zstd_df_find_kernel")) return rc;
        hipLaunchKernelGGL(aoclgpu::zstd_df_entropy_kernel, dim3(gridB < m ? gridB : m), dim3(64), ldsB, s, *b, ws, ldsB, f0, m, ctr + 1, (uint32_t*)(ws + metaOff));
        if (int rc = aoclgpu::check_launch("zstd_df_entropy_kernel")) return rc;
    }
    return AOCLGPU_OK;
}
}  // namespace

extern "C" int aoclgpu_zstd_last_split_slabs(void) { return g_df_last_slabs.load(); }

extern "C" int aoclgpu_zstd_compress_chunks(const aoclgpu_batch* b, void* stream)
{
    if (int rc = aoclgpu::check_batch(b, false)) return rc;
    if (b->nchunks == 0) return AOCLGPU_OK;
    hipStream_t s = (hipStream_t)stream;
    // launch classes: uniform mode -> body chunks share one LDS size, the ragged tail chunk gets its own launch;
    // explicit mode -> sizes live on the device, so every block gets the largest table (64 KiB)
    struct Launch { uint32_t first, count, lds; } L[2]; int nl = 0;
    const int level = (int)(((b->reserved >> 16) & 7u) ? ((b->reserved >> 16) & 7u) : ((b->reserved & AOCLGPU_FLAG_ZSTD_LEVEL2) ? 2u : 1u));
    if (level > 4) { aoclgpu::set_error_msg("zstd: levels 1-4 have a GPU match finder"); return AOCLGPU_ERR_ARG; }
    // double-fast (levels 3-4): the two hash tables of a resident workgroup live in the workspace, behind the sequence stores
    const size_t tabBytes = level >= 4 ? ((size_t)2 << 17) * 4 : (level == 3 ? (((size_t)1 << 16) + ((size_t)1 << 15)) * 4 : 0);
    if (b->src_off) { L[nl++] = { 0, b->nchunks, level >= 2 ? lds_for_chunk(128u << 10, level) : 65536u }; }
    else {
        const uint64_t tail = b->total_bytes - (uint64_t)(b->nchunks - 1) * b->chunk_bytes;
        if (b->chunk_bytes > (128u << 10) && b->nchunks > 1) { aoclgpu::set_error_msg("zstd: chunk_bytes > 128 KiB"); return AOCLGPU_ERR_ARG; }
        // a separate launch for the ragged tail chunk only when it needs more LDS than the body (a lone chunk is slow)
        if (b->nchunks > 1 && lds_for_chunk(tail, level) > lds_for_chunk(b->chunk_bytes, level)) {
            L[nl++] = { b->nchunks - 1, 1, lds_for_chunk(tail, level) };
            L[nl++] = { 0, b->nchunks - 1, lds_for_chunk(b->chunk_bytes, level) };
        } else L[nl++] = { 0, b->nchunks, lds_for_chunk(b->nchunks > 1 ? b->chunk_bytes : tail, level) };
    }
    const void* kfn = level >= 3 ? (const void*)aoclgpu::zstd_compress_kernel<true> : (const void*)aoclgpu::zstd_compress_kernel<false>;
    const uint32_t wpb = level >= 3 ? aoclgpu::kDfWaves : 1u;   // wavefronts (= chunks in flight) per workgroup
    aoclgpu::WsGuard guard;                               // held until the kernels that use the scratch are enqueued
    for (int i = 0; i < nl; ++i) {
        if (level >= 3) {
            const int rc = df_split_launch(b, L[i].first, L[i].count, tabBytes, guard, s);
            if (rc != 1) { if (rc != AOCLGPU_OK) return rc; continue; }   // 1: not worth it / no memory - the fused kernel below
        }
        // persistent blocks: exactly as many as are resident at once for this LDS size (gfx950 allocates LDS in 1280-byte
        // granules, the occupancy query counts bytes: the smaller of the two, see aoclgpu_zstd_decompress_chunks)
        int perCU = 0, dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kfn, (int)(64u * wpb), L[i].lds) != hipSuccess || perCU < 1) {
            aoclgpu::set_error_msg("zstd: occupancy query failed"); return AOCLGPU_ERR_HIP;
        }
        {
            const int byLds = (int)((160u << 10) / (((L[i].lds + 1279u) / 1280u) * 1280u));
            if (byLds >= 1 && byLds < perCU) perCU = byLds;
        }
        uint32_t grid = (uint32_t)(perCU * prop.multiProcessorCount);
        if (grid > (L[i].count + wpb - 1) / wpb) grid = (L[i].count + wpb - 1) / wpb;
        const size_t units = (size_t)grid * wpb;           // resident chunks: one workspace and (double-fast) one pair of tables each
        const size_t seqBytes = (units * aoclgpu::zstd_workspace_bytes() + 255) & ~(size_t)255;
        // (one hand-out counter per launch of this call, behind the scratch: the launches of a call run back to back on the stream)
        const size_t ctrOff = (seqBytes + units * tabBytes + 255) & ~(size_t)255;
        void* ws = get_workspace(guard, ctrOff + 256, stream);
        if (!ws) { aoclgpu::set_error_msg("zstd: workspace allocation failed"); return AOCLGPU_ERR_HIP; }
        uint8_t* tabBase = tabBytes ? (uint8_t*)ws + seqBytes : nullptr;
        uint32_t* counter = (uint32_t*)((uint8_t*)ws + ctrOff);
        if (hipMemsetAsync(counter, 0, 4, s) != hipSuccess) { aoclgpu::set_error_msg("zstd: hipMemsetAsync failed"); return AOCLGPU_ERR_HIP; }
        if (L[i].lds > 65536u) {
            hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L[i].lds);
            if (e != hipSuccess) { aoclgpu::set_error("hipFuncSetAttribute", e); return AOCLGPU_ERR_HIP; }
        }
        if (level >= 3) hipLaunchKernelGGL(aoclgpu::zstd_compress_kernel<true>, dim3(grid), dim3(64 * wpb), L[i].lds, s, *b, (uint8_t*)ws, L[i].lds, L[i].first, L[i].count,
                                           tabBase, (uint32_t)tabBytes, counter);
        else hipLaunchKernelGGL(aoclgpu::zstd_compress_kernel<false>, dim3(grid), dim3(64), L[i].lds, s, *b, (uint8_t*)ws, L[i].lds, L[i].first, L[i].count,
                                tabBase, (uint32_t)tabBytes, counter);
        if (int rc = aoclgpu::check_launch("zstd_compress_kernel")) return rc;
    }
    return AOCLGPU_OK;
}

extern "C" int aoclgpu_zstd_decompress_chunks(const aoclgpu_batch* b, void* stream)
{
    if (int rc = aoclgpu::check_batch(b, true)) return rc;
    if (b->nchunks == 0) return AOCLGPU_OK;
    aoclgpu::WsGuard guard;                               // also serialises the one-time occupancy query below
    // persistent blocks: exactly as many as are resident at once (LDS / VGPR limited), asked from the runtime
    static int perCU = 0, nCU = 0;
    if (!perCU) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, (const void*)aoclgpu::zstd_decompress_kernel, 64, 0) != hipSuccess || perCU < 1) {
            aoclgpu::set_error_msg("zstd: occupancy query failed"); perCU = 0; return AOCLGPU_ERR_HIP;
        }
        nCU = prop.multiProcessorCount;
        // The occupancy query counts LDS by the byte; gfx950 hands its 160 KB out in 1280-byte granules (128 per CU: with
        // 13.2 KB per workgroup 11 fit, not the 12 the query reports, and 17.5 KB x 9 fits exactly).  One workgroup too many
        // per CU means a second round of workgroups that start when the first have done all their chunks: 38.9 instead of
        // 58.1 GiB/s.
        const int byLds = (int)((160u << 10) / (((aoclgpu::zstd_decoder_lds_bytes() + 1279u) / 1280u) * 1280u));
        if (byLds >= 1 && byLds < perCU) perCU = byLds;
#ifdef AOCLGPU_EXPERIMENTS
        if (const char* g = getenv("AOCLGPU_ZSTD_DEC_PER_CU")) perCU = atoi(g) > 0 ? atoi(g) : perCU;      // tuning hook
#endif
    }
    uint32_t grid = (uint32_t)(perCU * nCU);
    if (grid > b->nchunks) grid = b->nchunks;
    // the decoder's literal buffers share the stream's workspace with the compressor's (same stream order)
    const size_t ctrOff = ((size_t)grid * aoclgpu::kDecWs + 255) & ~(size_t)255;
    void* ws = get_workspace(guard, ctrOff + 256, stream);
    if (!ws) { aoclgpu::set_error_msg("zstd: workspace allocation failed"); return AOCLGPU_ERR_HIP; }
    uint32_t* counter = (uint32_t*)((uint8_t*)ws + ctrOff);
    if (hipMemsetAsync(counter, 0, 4, (hipStream_t)stream) != hipSuccess) { aoclgpu::set_error_msg("zstd: hipMemsetAsync failed"); return AOCLGPU_ERR_HIP; }
    hipLaunchKernelGGL(aoclgpu::zstd_decompress_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, *b, (uint8_t*)ws, counter);
    return aoclgpu::check_launch("zstd_decompress_kernel");
}

// One frame per input, any size below 2 GiB (see zstd_frame_wave).
extern "C" int aoclgpu_zstd_compress_frames(const aoclgpu_zstd_frames* f, void* stream)
{
    if (!f || (f->nframes && (!f->src_off || !f->dst_off || !f->out_len || !f->dst))) { aoclgpu::set_error_msg("zstd frames: bad arguments"); return AOCLGPU_ERR_ARG; }
    const bool df = f->strategy == 2;
    if ((f->strategy != 1 && !df) || f->windowLog < 10 || f->windowLog > 31 || f->hashLog < 6 || f->hashLog > 24 || f->minMatch < 4 || f->minMatch > 7 ||
        (df ? (f->chainLog < 6 || f->chainLog > 24) : f->targetLength > (1u << 17))) {
        aoclgpu::set_error_msg("zstd frames: parameters outside the GPU parses (strategy 1 fast / 2 double-fast, windowLog 10-31, hashLog 6-24, "
                               "chainLog 6-24 for double-fast, minMatch 4-7, targetLength 0-131072 for fast)");
        return AOCLGPU_ERR_ARG;
    }
    if (f->nframes == 0) return AOCLGPU_OK;
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { aoclgpu::set_error_msg("zstd frames: device query failed"); return AOCLGPU_ERR_HIP; }
    // Two finders for the fast strategy, same bytes: table + source ring in LDS with chain batches (one frame per CU, ~2.4x
    // the speed per frame) for a lone frame or a small batch, table in HBM (9 frames per CU) for the bulk.
    const bool canLds = !df && f->hashLog <= 14;
    const bool useLds = (f->flags & AOCLGPU_ZSTD_FRAMES_FINDER_LDS) ? canLds
                      : ((f->flags & AOCLGPU_ZSTD_FRAMES_FINDER_HBM) ? false : (canLds && f->nframes <= 2u * (uint32_t)prop.multiProcessorCount));
    if ((f->flags & AOCLGPU_ZSTD_FRAMES_FINDER_LDS) && !canLds) { aoclgpu::set_error_msg("zstd frames: the LDS finder serves the fast strategy with hashLog <= 14"); return AOCLGPU_ERR_ARG; }
    // the fast parser's step: targetLength + !targetLength + 1 (zstd_fast.c:440); above 2 (levels below -1) the pair schedule is
    // built on the device in front of the frames
    const bool stepped = !df && f->targetLength > 1;
    aoclgpu::WsGuard guard;
    if (useLds) {
        const uint32_t ldsBytes = (uint32_t)(((size_t)4 << f->hashLog) + aoclgpu::kSrcRingBytes + aoclgpu::zstd_frame_lds_bytes());
        uint32_t grid = (uint32_t)prop.multiProcessorCount * ((160u << 10) / ldsBytes ? (160u << 10) / (((ldsBytes + 1279u) / 1280u) * 1280u) : 1u);
        if (grid > f->nframes) grid = f->nframes;
        const size_t wsBytes = ((size_t)grid * aoclgpu::zstd_workspace_bytes() + 255) & ~(size_t)255;
        void* ws = get_workspace(guard, wsBytes + (stepped ? aoclgpu::kFastSchedBytes : 0), stream);
        if (!ws) { aoclgpu::set_error_msg("zstd frames: workspace allocation failed"); return AOCLGPU_ERR_HIP; }
        uint32_t* sched = stepped ? (uint32_t*)((uint8_t*)ws + wsBytes) : nullptr;
        if (stepped) hipLaunchKernelGGL(aoclgpu::zstd_fast_sched_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sched, f->targetLength + 1u);
        if (ldsBytes > 65536u) {
            hipError_t e = hipFuncSetAttribute((const void*)aoclgpu::zstd_frame_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
            if (e != hipSuccess) { aoclgpu::set_error("hipFuncSetAttribute", e); return AOCLGPU_ERR_HIP; }
        }
        hipLaunchKernelGGL(aoclgpu::zstd_frame_lds_kernel, dim3(grid), dim3(64), ldsBytes, (hipStream_t)stream, *f, (uint8_t*)ws, (const uint32_t*)sched);
        return aoclgpu::check_launch("zstd_frame_lds_kernel");
    }
    // one wave per frame; as many frames per CU as its LDS holds (granules of 1280 bytes, 128 per CU), at most 12 (163 VGPRs)
    uint32_t perCU = 128u / (uint32_t)((aoclgpu::zstd_frame_lds_bytes() + 1279) / 1280);
    if (perCU > 12) perCU = 12;
    uint32_t grid = (uint32_t)prop.multiProcessorCount * perCU;
    if (grid > f->nframes) grid = f->nframes;
    const size_t tabBytes = ((size_t)4 << f->hashLog) + (df ? (size_t)4 << f->chainLog : 0);
    if ((size_t)grid * tabBytes > ((size_t)1 << 30)) grid = (uint32_t)(((size_t)1 << 30) / tabBytes);   // at most 1 GiB of tables
    const size_t seqBytes = ((size_t)grid * aoclgpu::zstd_workspace_bytes() + 255) & ~(size_t)255;
    const size_t allTabs = ((size_t)grid * tabBytes + 255) & ~(size_t)255;
    void* ws = get_workspace(guard, seqBytes + allTabs + (stepped ? aoclgpu::kFastSchedBytes : 0), stream);
    if (!ws) { aoclgpu::set_error_msg("zstd frames: workspace allocation failed"); return AOCLGPU_ERR_HIP; }
    uint32_t* sched = stepped ? (uint32_t*)((uint8_t*)ws + seqBytes + allTabs) : nullptr;
    if (stepped) hipLaunchKernelGGL(aoclgpu::zstd_fast_sched_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sched, f->targetLength + 1u);
    hipLaunchKernelGGL(aoclgpu::zstd_frame_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, *f, (uint8_t*)ws, (uint8_t*)ws + seqBytes, (uint32_t)tabBytes, (const uint32_t*)sched);
    return aoclgpu::check_launch("zstd_frame_kernel");
}
