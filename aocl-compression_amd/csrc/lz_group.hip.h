// csrc/lz_group.hip.h -- the decoders' group execution (LZ4 blocks, Snappy elements).
//
// A decoder scans its token chain into small records and then executes up to 64 of them at once, one per lane: lane i
// holds a literal run (lit bytes at src + litsrc) followed by a copy of earlier output (mlen bytes, `off` back); prefix
// sums have given it its place (lpos, mst = lpos + lit) in a 2 KB output block that lives in LDS behind 64 bytes of
// history (the output just in front of the block).  lz_group_execute does the rest:
//   1. literals (input -> block) and copies whose source ends in front of the block (output -> block): four pieces per size
//      class at 0, k, 2k and len - k (clamped) - overlapping pieces carry the same bytes, every access stays inside its
//      run - with every load requested before the first write: one trip to memory for the whole group; runs of 65 .. 255
//      bytes (rare) follow by the whole wave, 4 bytes per lane;
//   2. copies that read the block itself (or the history): in record order as LDS-to-LDS copies, a byte per lane;
//   3. the block leaves with 16-byte stores and its last 64 bytes become the next block's history.
// The caller has checked offsets and capacities: act lanes are [0, ng), every copy source starts at or after the start of
// the output and - when it reaches into the block - not further in front of it than the history.
#pragma once
#include "common.hip.h"

namespace aoclgpu {

constexpr uint32_t kGrpBlock = 2048, kGrpHist = 64;

// One round of a group's loads (SPLIT only): the pieces of a run as global loads with a scalar base and a 32-bit lane offset,
// each under the lane mask of its size class, AND the wait for them, in ONE asm statement.  (The compiler moves the 64-bit address
// arithmetic of a conditional load out of its block and then cannot fold base + offset into the instruction: every piece cost a
// 64-bit vector add and a register pair that lived from there to the load - in the Zstd decoder, where the budget is 80 registers,
// those pairs were spilled.  Until round 6 every load was its own asm statement and the wait another one: the compiler does not
// count loads it cannot see, so nothing kept it from copying or spilling a destination register between a load and the wait.
// Here the destinations are only defined when the statement ends, behind s_waitcnt.)  Lanes outside a piece's mask keep
// whatever the register held: the caller writes a piece under the same mask.  exec is saved and restored.
//   a0: offset of the run; pieces at a0, a0 + 16, a0 + 32, aE16 (16-byte class k16*), a0, a0 + 4, a0 + 8, aE4 (4-byte class k4*),
//   a0, aMid, aLast (1 .. 3 bytes, k1).
__device__ __forceinline__ void grp_load_round(const gbyte* sbase, const uint32_t a0, const uint32_t aE16, const uint32_t aE4, const uint32_t aMid, const uint32_t aLast,
                                               const uint64_t k16, const uint64_t k16b, const uint64_t k16c, const uint64_t k16d,
                                               const uint64_t k4, const uint64_t k4b, const uint64_t k4c, const uint64_t k4d, const uint64_t k1,
                                               u32x4& q0, u32x4& q1, u32x4& q2, u32x4& q3, uint32_t& r0, uint32_t& r1, uint32_t& r2, uint32_t& r3,
                                               uint32_t& b0, uint32_t& b1, uint32_t& b2)
{
    uint64_t sv;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "s_mov_b64 exec, %[k16]\n\t"  "global_load_dwordx4 %[q0], %[a0], %[sb]\n\t"
                 "s_mov_b64 exec, %[k16b]\n\t" "global_load_dwordx4 %[q1], %[a0], %[sb] offset:16\n\t"
                 "s_mov_b64 exec, %[k16c]\n\t" "global_load_dwordx4 %[q2], %[a0], %[sb] offset:32\n\t"
                 "s_mov_b64 exec, %[k16d]\n\t" "global_load_dwordx4 %[q3], %[aE16], %[sb]\n\t"
                 "s_mov_b64 exec, %[k4]\n\t"   "global_load_dword %[r0], %[a0], %[sb]\n\t"
                 "s_mov_b64 exec, %[k4b]\n\t"  "global_load_dword %[r1], %[a0], %[sb] offset:4\n\t"
                 "s_mov_b64 exec, %[k4c]\n\t"  "global_load_dword %[r2], %[a0], %[sb] offset:8\n\t"
                 "s_mov_b64 exec, %[k4d]\n\t"  "global_load_dword %[r3], %[aE4], %[sb]\n\t"
                 "s_mov_b64 exec, %[k1]\n\t"   "global_load_ubyte %[b0], %[a0], %[sb]\n\t"
                                                "global_load_ubyte %[b1], %[aMid], %[sb]\n\t"
                                                "global_load_ubyte %[b2], %[aLast], %[sb]\n\t"
                 "s_mov_b64 exec, %[sv]\n\t"
                 "s_waitcnt vmcnt(0)"
                 : [sv] "=&s"(sv), [q0] "=&v"(q0), [q1] "=&v"(q1), [q2] "=&v"(q2), [q3] "=&v"(q3), [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3),
                   [b0] "=&v"(b0), [b1] "=&v"(b1), [b2] "=&v"(b2)
                 : [sb] "s"(sbase), [a0] "v"(a0), [aE16] "v"(aE16), [aE4] "v"(aE4), [aMid] "v"(aMid), [aLast] "v"(aLast),
                   [k16] "s"(k16), [k16b] "s"(k16b), [k16c] "s"(k16c), [k16d] "s"(k16d), [k4] "s"(k4), [k4b] "s"(k4b), [k4c] "s"(k4c), [k4d] "s"(k4d), [k1] "s"(k1)
                 : "memory");
}
constexpr uint32_t kGrpLds = kGrpHist + kGrpBlock + 16;      // cb: history | block | slack for the history read behind a full block

// The in-block copies of a group, in record order, while they are SHORT AND PLAIN (at most 63 bytes, offset >= length: one pass,
// no lane reads what the pass writes; `fast` = the lanes whose copy is one) - hand-written: a copy is a round trip through LDS
// that the whole wave sits through, there are ten of them in an average group (a timing-only build without them: 13.3 instead of
// 17.9 ms for the LZ4 bench), and the compiler's loop spends ~40 instructions and five taken branches on each.  Here a copy is
// exec = its lanes (s_bfm), read, write at read + offset; while the read is on its way the NEXT copy is looked up (find the lane,
// three readlanes; the loop is unrolled twice so that the two sets of scalars swap roles without moves), so a copy costs ~17
// instructions, half a taken branch and no exposed LDS latency.  The LDS unit executes a
// wave's instructions in issue order: the next copy's read sees this copy's write without a wait in between.  Needs all 64
// lanes active on entry.  Returns with m = the copies still to do: empty, or its lowest lane is a copy for the general path.
__device__ __forceinline__ void lz_near_fast(uint64_t& m, const uint64_t fast, const uint32_t mlen, const uint32_t off, const uint32_t spA, const uint32_t L)
{
    uint32_t k, ml, of, sp, ml2, of2, sp2, va, vb;
    uint64_t sv;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "s_cmp_eq_u64 %0, 0\n\t"
                 "s_cbranch_scc1 9f\n\t"
                 "s_ff1_i32_b64 %1, %0\n\t"
                 "s_bitcmp0_b64 %11, %1\n\t"
                 "s_cbranch_scc1 9f\n\t"
                 "v_readlane_b32 %2, %12, %1\n\t"
                 "v_readlane_b32 %3, %13, %1\n\t"
                 "v_readlane_b32 %4, %14, %1\n"
                 "1:\n\t"
                 "s_bfm_b64 exec, %2, 0\n\t"
                 "v_add_u32_e32 %8, %4, %15\n\t"
                 "ds_read_u8 %9, %8\n\t"
                 "s_bitset0_b64 %0, %1\n\t"
                 "s_cmp_eq_u64 %0, 0\n\t"
                 "s_cbranch_scc1 3f\n\t"
                 "s_ff1_i32_b64 %1, %0\n\t"
                 "s_bitcmp0_b64 %11, %1\n\t"
                 "s_cbranch_scc1 3f\n\t"
                 "v_readlane_b32 %5, %12, %1\n\t"
                 "v_readlane_b32 %6, %13, %1\n\t"
                 "v_readlane_b32 %7, %14, %1\n\t"
                 "v_add_u32_e32 %8, %3, %8\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "ds_write_b8 %8, %9\n\t"
                 // second half: the two register sets have swapped roles
                 "s_bfm_b64 exec, %5, 0\n\t"
                 "v_add_u32_e32 %8, %7, %15\n\t"
                 "ds_read_u8 %9, %8\n\t"
                 "s_bitset0_b64 %0, %1\n\t"
                 "s_cmp_eq_u64 %0, 0\n\t"
                 "s_cbranch_scc1 4f\n\t"
                 "s_ff1_i32_b64 %1, %0\n\t"
                 "s_bitcmp0_b64 %11, %1\n\t"
                 "s_cbranch_scc1 4f\n\t"
                 "v_readlane_b32 %2, %12, %1\n\t"
                 "v_readlane_b32 %3, %13, %1\n\t"
                 "v_readlane_b32 %4, %14, %1\n\t"
                 "v_add_u32_e32 %8, %6, %8\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "ds_write_b8 %8, %9\n\t"
                 "s_branch 1b\n"
                 "3:\n\t"
                 "v_add_u32_e32 %8, %3, %8\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "ds_write_b8 %8, %9\n\t"
                 "s_branch 9f\n"
                 "4:\n\t"
                 "v_add_u32_e32 %8, %6, %8\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "ds_write_b8 %8, %9\n"
                 "9:\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : "+s"(m), "=&s"(k), "=&s"(ml), "=&s"(of), "=&s"(sp), "=&s"(ml2), "=&s"(of2), "=&s"(sp2), "=&v"(va), "=&v"(vb), [sv] "=&s"(sv)
                 : "s"(fast), "v"(mlen), "v"(off), "v"(spA), "v"(L)
                 : "scc", "memory");
}

// TINY: copies shorter than 4 bytes exist (Snappy, Zstd).  SPLIT: literals and far copies in two rounds of loads instead of one (half
// the registers in flight).  Returns the group's output bytes (= what `op` advances by).
template <bool TINY, bool SPLIT = false>
__device__ __forceinline__ uint32_t lz_group_execute(const gbyte* __restrict__ src, gbyte* dst, lbyte* cb, bool& hist_ok, const uint32_t op,
                                                     const uint32_t ng, const bool act, const uint32_t lit, const uint32_t mlen, const uint32_t off,
                                                     const uint32_t litsrc, const uint32_t lpos, const uint32_t mst, const uint32_t total)
{
    typedef __attribute__((address_space(3))) uint32_t L32;
    typedef u32x4 __attribute__((aligned(1))) U128;
    typedef __attribute__((address_space(1))) U128 G128;
    typedef __attribute__((address_space(3))) U128 S128;
    typedef uint32_t __attribute__((aligned(1))) U32;
    typedef __attribute__((address_space(3))) U32 S32;
    const uint32_t L = (uint32_t)lane_id();
    if (!hist_ok) {                                           // the 64 bytes in front of the block (after a sequential step, at the start)
        uint8_t b = 0;
        if (op + L >= kGrpHist) b = dst[op + L - kGrpHist];
        cb[L] = b;
    }
    lbyte* blk = cb + kGrpHist;
    const int32_t srel = (int32_t)mst - (int32_t)off;                          // source start, block-relative
    const bool far = act && mlen && srel + (int32_t)mlen <= 0;
    const bool near = act && mlen && !far;
    const uint32_t mo = (uint32_t)((int32_t)op + srel);                        // far copies: source start in dst
    const bool l16 = act && lit >= 16 && lit <= 64, l4 = act && lit >= 4 && lit < 16, l1 = act && lit >= 1 && lit < 4;
    const bool m16 = far && mlen >= 16 && mlen <= 64, m4 = far && mlen >= 4 && mlen < 16, m1 = TINY && far && mlen < 4;
    // Pieces of a run: the first at 0, the last at len - k, and between them only what the length needs (at k and 2k) - a lane
    // that does not need a piece sits the instruction out: an LDS (and a memory) instruction costs by its active lanes, and most
    // runs of a class are in its lower half (until late in round 5 all four pieces were always moved, clamped onto each other:
    // the LZ4 decoder's LDS pipe was 80 % busy, SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES).
    const uint32_t le16 = lit - 16, le4 = lit - 4, me16 = mlen - 16, me4 = mlen - 4;
    const bool l16b = l16 && lit > 32, l16c = l16 && lit > 48, l16d = l16 && lit > 16;
    const bool m16b = m16 && mlen > 32, m16c = m16 && mlen > 48, m16d = m16 && mlen > 16;
    const bool l4b = l4 && lit > 8, l4c = l4 && lit > 12, l4d = l4 && lit > 4;
    const bool m4b = m4 && mlen > 8, m4c = m4 && mlen > 12, m4d = m4 && mlen > 4;
    lbyte* ld_ = blk + lpos; lbyte* md_ = blk + mst;
    if constexpr (SPLIT) {
        // two rounds - the literals, then the far copies - with half the registers in flight (the Zstd decoder's budget)
        {
            u32x4 q0, q1, q2, q3;
            uint32_t r0, r1, r2, r3, b0, b1, b2;
            grp_load_round(src, litsrc, litsrc + le16, litsrc + le4, litsrc + (lit >> 1), litsrc + lit - 1,
                           ballot(l16), ballot(l16b), ballot(l16c), ballot(l16d), ballot(l4), ballot(l4b), ballot(l4c), ballot(l4d), ballot(l1),
                           q0, q1, q2, q3, r0, r1, r2, r3, b0, b1, b2);
            if (l16) *(S128*)ld_ = q0;
            if (l16b) *(S128*)(ld_ + 16) = q1;
            if (l16c) *(S128*)(ld_ + 32) = q2;
            if (l16d) *(S128*)(ld_ + le16) = q3;
            if (l4) *(S32*)ld_ = r0;
            if (l4b) *(S32*)(ld_ + 4) = r1;
            if (l4c) *(S32*)(ld_ + 8) = r2;
            if (l4d) *(S32*)(ld_ + le4) = r3;
            if (l1) { ld_[0] = (uint8_t)b0; ld_[lit >> 1] = (uint8_t)b1; ld_[lit - 1] = (uint8_t)b2; }
        }
        asm volatile("" ::: "memory");
        {
            u32x4 g0, g1, g2, g3;
            uint32_t h0, h1, h2, h3, c0, c1, c2;
            grp_load_round(dst, mo, mo + me16, mo + me4, mo + (mlen >> 1), mo + mlen - 1,
                           ballot(m16), ballot(m16b), ballot(m16c), ballot(m16d), ballot(m4), ballot(m4b), ballot(m4c), ballot(m4d), ballot(m1),
                           g0, g1, g2, g3, h0, h1, h2, h3, c0, c1, c2);
            if (m16) *(S128*)md_ = g0;
            if (m16b) *(S128*)(md_ + 16) = g1;
            if (m16c) *(S128*)(md_ + 32) = g2;
            if (m16d) *(S128*)(md_ + me16) = g3;
            if (m4) *(S32*)md_ = h0;
            if (m4b) *(S32*)(md_ + 4) = h1;
            if (m4c) *(S32*)(md_ + 8) = h2;
            if (m4d) *(S32*)(md_ + me4) = h3;
            if (m1) { md_[0] = (uint8_t)c0; md_[mlen >> 1] = (uint8_t)c1; md_[mlen - 1] = (uint8_t)c2; }
        }
    } else {
    u32x4 q0 = {0u, 0u, 0u, 0u}, q1 = q0, q2 = q0, q3 = q0, g0 = q0, g1 = q0, g2 = q0, g3 = q0;
    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, h0 = 0, h1 = 0, h2 = 0, h3 = 0;
    uint8_t b0 = 0, b1 = 0, b2 = 0, c0 = 0, c1 = 0, c2 = 0;
    if (l16) q0 = *(const G128*)(src + litsrc);
    if (l16b) q1 = *(const G128*)(src + (litsrc + 16));
    if (l16c) q2 = *(const G128*)(src + (litsrc + 32));
    if (l16d) q3 = *(const G128*)(src + (litsrc + le16));
    if (m16) g0 = *(const G128*)(dst + mo);
    if (m16b) g1 = *(const G128*)(dst + (mo + 16));
    if (m16c) g2 = *(const G128*)(dst + (mo + 32));
    if (m16d) g3 = *(const G128*)(dst + (mo + me16));
    if (l4) r0 = ld32(src + litsrc);
    if (l4b) r1 = ld32(src + (litsrc + 4));
    if (l4c) r2 = ld32(src + (litsrc + 8));
    if (l4d) r3 = ld32(src + (litsrc + le4));
    if (m4) h0 = ld32(dst + mo);
    if (m4b) h1 = ld32(dst + (mo + 4));
    if (m4c) h2 = ld32(dst + (mo + 8));
    if (m4d) h3 = ld32(dst + (mo + me4));
    if (l1) { b0 = src[litsrc]; b1 = src[litsrc + (lit >> 1)]; b2 = src[litsrc + lit - 1]; }
    if (m1) { c0 = dst[mo]; c1 = dst[mo + (mlen >> 1)]; c2 = dst[mo + mlen - 1]; }
    asm volatile("" ::: "memory");
    if (l16) *(S128*)ld_ = q0;
    if (l16b) *(S128*)(ld_ + 16) = q1;
    if (l16c) *(S128*)(ld_ + 32) = q2;
    if (l16d) *(S128*)(ld_ + le16) = q3;
    if (m16) *(S128*)md_ = g0;
    if (m16b) *(S128*)(md_ + 16) = g1;
    if (m16c) *(S128*)(md_ + 32) = g2;
    if (m16d) *(S128*)(md_ + me16) = g3;
    if (l4) *(S32*)ld_ = r0;
    if (l4b) *(S32*)(ld_ + 4) = r1;
    if (l4c) *(S32*)(ld_ + 8) = r2;
    if (l4d) *(S32*)(ld_ + le4) = r3;
    if (m4) *(S32*)md_ = h0;
    if (m4b) *(S32*)(md_ + 4) = h1;
    if (m4c) *(S32*)(md_ + 8) = h2;
    if (m4d) *(S32*)(md_ + me4) = h3;
    if (l1) { ld_[0] = b0; ld_[lit >> 1] = b1; ld_[lit - 1] = b2; }
    if (m1) { md_[0] = c0; md_[mlen >> 1] = c1; md_[mlen - 1] = c2; }
    }
    // runs of 65 .. 259 bytes (rare): the whole wave, 4 bytes per lane, one after the other
    for (uint64_t m = ballot(act && lit > 64); m; m &= m - 1) {
        const int k = (int)__builtin_ctzll(m);
        const uint32_t n_ = lane_get(lit, k), so = lane_get(litsrc, k), d_ = lane_get(lpos, k);
        for (uint32_t o = 4u * L; o < n_; o += 4u * kWave) { const uint32_t o4 = o < n_ - 4 ? o : n_ - 4; *(S32*)(blk + d_ + o4) = ld32(src + (so + o4)); }
    }
    for (uint64_t m = ballot(far && mlen > 64); m; m &= m - 1) {
        const int k = (int)__builtin_ctzll(m);
        const uint32_t n_ = lane_get(mlen, k), so = lane_get(mo, k), d_ = lane_get(mst, k);
        for (uint32_t o = 4u * L; o < n_; o += 4u * kWave) { const uint32_t o4 = o < n_ - 4 ? o : n_ - 4; *(S32*)(blk + d_ + o4) = ld32(dst + (so + o4)); }   // (a match may be 259 bytes: one lane comes round again)
    }
    wave_fence();
    // copies that read the block (or the history in front of it): in record order, the whole wave a byte per lane - the short
    // plain ones by lz_near_fast, a long copy or a pattern here
    {
        const uint32_t blkA = (uint32_t)(uintptr_t)blk;
        const uint32_t spA = blkA + (uint32_t)srel;              // (a copy's destination is its source + offset)
        uint64_t m = ballot(near);
        const uint64_t fast = ballot(near && mlen <= 63 && off >= mlen);
        while (m) {
#ifndef LZ_GROUP_NEAR_CXX
            lz_near_fast(m, fast, mlen, off, spA, L);
            if (!m) break;
#endif
            const int k = (int)__builtin_ctzll(m);
            const uint32_t ml = lane_get(mlen, k), of = lane_get(off, k);
            lbyte* sp = blk + (int32_t)lane_get((uint32_t)srel, k); lbyte* dp = blk + lane_get(mst, k);
            if (of >= 64 || of >= ml) {                      // no lane reads what another lane of the same pass writes
                for (uint32_t i = L; i < ml; i += kWave) { const uint8_t b = sp[i]; dp[i] = b; wave_fence(); }   // (a pass reads what the pass before wrote)
            } else {                                          // offset < length: the pattern repeats
                for (uint32_t i = L; i < ml; i += kWave) { const uint8_t b = sp[i % of]; dp[i] = b; }
            }
            wave_fence();
            m &= m - 1;
        }
    }
    // the block leaves: 16 bytes per lane, then the odd bytes
    const uint32_t full = total & ~15u;
    for (uint32_t o = 16u * L; o < full; o += 16u * kWave) *(G128*)(dst + (op + o)) = *(const S128*)(blk + o);
    if (L < (total & 15u)) dst[op + full + L] = blk[full + L];
    // the last 64 bytes become the next block's history
    {
        uint32_t hv = 0;
        if (L < 16) hv = *(const S32*)(cb + total + 4u * L);
        wave_fence();
        if (L < 16) *(L32*)(cb + 4u * L) = hv;
    }
    hist_ok = true;
    wave_fence();
    return total;
}

// The scan's walk over the token starts of a round of 64 byte positions: nx[k] = lane of the token that follows a token at
// lane k, or negative when the scan does not take that token.  mask |= 1 << cur; cur = nx[cur] until cur >= 64 (stop = 0)
// or a token that is not taken (stop = 2, cur stays on it).
// Round 5: four tokens per trip of the loop.  A negative link is >= 64 as an unsigned number, so ONE compare ends the chain for
// both reasons and which one it was is looked at behind the loop; the exits are forward branches that are not taken until the
// chain ends, the only taken branch is the back edge every fourth token (a taken branch costs a wave an instruction-buffer
// refill, the old loop took one per token): readlane, compare, branch, bit set = 4 scalar instructions per token instead of 7.
// (s_nop: a v_readlane's lane select written by a VALU instruction - the readlane before it - needs four wait states.)
__device__ __forceinline__ void lz_walk_tokens(const uint32_t nx, uint64_t& mask, uint32_t& cur, uint32_t& stop, const uint32_t first = 0)
{   // first: the lane the walk starts at (< 64)
    uint32_t t;
#ifdef LZ_WALK_OLD
    asm volatile("s_mov_b64 %0, 0\n\t"
                 "s_mov_b32 %1, %5\n\t"
                 "s_mov_b32 %2, 0\n"
                 "1:\n\t"
                 "v_readlane_b32 %3, %4, %1\n\t"
                 "s_cmp_lt_i32 %3, 0\n\t"
                 "s_cbranch_scc1 2f\n\t"
                 "s_bitset1_b64 %0, %1\n\t"
                 "s_mov_b32 %1, %3\n\t"
                 "s_cmp_lt_u32 %1, 64\n\t"
                 "s_cbranch_scc1 1b\n\t"
                 "s_branch 3f\n"
                 "2:\n\t"
                 "s_mov_b32 %2, 2\n"
                 "3:"
                 : "=&s"(mask), "=&s"(cur), "=&s"(stop), "=&s"(t) : "v"(nx), "s"(first) : "scc");
#else
#define LZ_WALK_PAIR \
                 "v_readlane_b32 %3, %4, %1\n\t" \
                 "s_cmp_lt_u32 %3, 64\n\t" \
                 "s_cbranch_scc0 4f\n\t" \
                 "s_bitset1_b64 %0, %1\n\t" \
                 "s_nop 0\n\t" \
                 "v_readlane_b32 %1, %4, %3\n\t" \
                 "s_cmp_lt_u32 %1, 64\n\t" \
                 "s_cbranch_scc0 5f\n\t" \
                 "s_bitset1_b64 %0, %3\n\t"
    asm volatile("s_mov_b64 %0, 0\n\t"
                 "s_mov_b32 %1, %5\n\t"
                 "s_mov_b32 %2, 0\n"
                 "1:\n\t"
                 LZ_WALK_PAIR
                 "s_nop 0\n\t"
                 LZ_WALK_PAIR
                 "s_branch 1b\n"
                 "4:\n\t"                                   // the token at %1 links to %3: the end of the round or a token the scan does not take
                 "s_cmp_lt_i32 %3, 0\n\t"
                 "s_cbranch_scc1 6f\n\t"
                 "s_bitset1_b64 %0, %1\n\t"
                 "s_mov_b32 %1, %3\n\t"
                 "s_branch 7f\n"
                 "5:\n\t"                                   // the token at %3 links to %1
                 "s_cmp_lt_i32 %1, 0\n\t"
                 "s_cbranch_scc1 8f\n\t"
                 "s_bitset1_b64 %0, %3\n\t"
                 "s_branch 7f\n"
                 "8:\n\t"
                 "s_mov_b32 %1, %3\n"
                 "6:\n\t"
                 "s_mov_b32 %2, 2\n"
                 "7:"
                 : "=&s"(mask), "=&s"(cur), "=&s"(stop), "=&s"(t) : "v"(nx), "s"(first) : "scc");
#undef LZ_WALK_PAIR
#endif
}

}  // namespace aoclgpu
