// tests/zstd_chain_model.cpp -- TEST INFRASTRUCTURE: lane-by-lane CPU model of the Zstd match-finder kernel
// (aocl-compression_amd/csrc/zstd_kernels.hip, fast_block) that runs the kernel's own scalar chain walk
// (csrc/zstd_chain.h) on the host.  tests/test_zstd_chain_model.py compares its sequences with the oracle's sequential
// parser (oracle/zstd_oracle.c, fast_block) on many inputs, so the batching logic - which lanes are trustworthy, which
// table writes are committed, when a chain has to stop - is validated without a GPU.
// Mode bits (useChain): 1 chain batches, 2 full statistics, 4 resumed batches, 8 resume only in front of a matching lane,
// 16 long matches inside the batch, 32 duplicate lanes (a slot named by an earlier lane of the batch) do not cut the batch:
// the walk stops at them and they are probed again with the writes made since (the kernel runs 1 | 4 | 8; 16 and 32 were
// measured on the GPU and rejected - tools/experiments/README.md - and stay here as the statement of their rules).
//
// Build: g++ -O2 -shared -fPIC -I aocl-compression_amd/csrc tests/zstd_chain_model.cpp -o tests/_build/libzchain_model.so
#include "zstd_chain.h"
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {
typedef uint8_t u8; typedef uint32_t u32; typedef uint64_t u64;
u32 rd32(const u8* p) { u32 v; memcpy(&v, p, 4); return v; }
u64 rd64(const u8* p) { u64 v; memcpy(&v, p, 8); return v; }
u32 zhash(const u8* p, u32 hlog, u32 mls)
{
    const u64 v = rd64(p);
    if (mls == 4) return ((u32)v * 2654435761u) >> (32 - hlog);
    if (mls == 5) return (u32)(((v << 24) * 889523592379ull) >> (64 - hlog));
    if (mls == 6) return (u32)(((v << 16) * 227718039650203ull) >> (64 - hlog));
    return (u32)(((v << 8) * 58295818150454627ull) >> (64 - hlog));
}
u32 zcount(const u8* s, u32 a, u32 b, u32 end) { u32 n = 0; while (a + n < end && s[a + n] == s[b + n]) n++; return n; }
int hb(u32 v) { int r = 0; while (v >>= 1) r++; return r; }

// pair schedule (same construction as FastSched in zstd_kernels.hip)
struct Sched {
    enum { N = 70000 };
    std::vector<u32> a, sigma, st;
    Sched() : a(N), sigma(N), st(N) {
        u32 step = 2, nextStep = 128, A = 0, sig = 2;
        for (u32 m = 0; m < N; ++m) {
            a[m] = A; sigma[m] = sig; st[m] = step;
            const u32 A2 = A + sig; sig = step;
            if (A2 + step >= nextStep) { step++; nextStep += 128; }
            A = A2;
        }
    }
};
const Sched g_sched;
u64* g_trace = nullptr; u32 g_trace_n = 0;      // optional per-batch trace (debugging the kernel against the model)

struct Seq { u32 ll, off, ml; };

struct Model {
    const u8* src; u32 n, hlog, mls;
    std::vector<u32> table;
    std::vector<Seq> seqs;
    u64 litSize = 0;
    u32 lastLits = 0;
    int resumeMargin = 8;
    u64 stat[40] = {0};   // batches, fast sequences, specials, restarts, continues, sequential searches, tails, post checks; [8+why] specials by reason
    void put(u32 pos) { table[zhash(src + pos, hlog, mls)] = pos + 2; }
};

}  // namespace

extern "C" void zchain_model_trace(u64* buf) { g_trace = buf; g_trace_n = 0; }
extern "C" u32 zchain_model_trace_count(void) { return g_trace_n; }

struct Win { u32 dictLimit, lowLimit, maxDist; };
static u32 lowest_prefix(const Win& w, u32 curr) { return (curr - w.dictLimit > w.maxDist) ? curr - w.maxDist : w.dictLimit; }

// one block [bs, n) of the frame at M.src: the kernel's fast_block<.., FRAME> lane by lane (a chunk is the frame's only block)
static void model_block(Model& M, const u32 bs, const u32 n, const Win& win, u32& rep1io, u32& rep2io, int useChain)
{
    const u8* src = M.src;
    const u32 prefixIdx = lowest_prefix(win, n + 2), prefixPos = prefixIdx - 2;
    u32 anchor = bs, s = bs + (bs == prefixPos ? 1u : 0u), rep1 = rep1io, rep2 = rep2io, saved1 = 0, saved2 = 0, i0 = 0;
    {   const u32 curr = s + 2, maxRep = curr - lowest_prefix(win, curr);
        if (rep2 > maxRep) { saved2 = rep2; rep2 = 0; }
        if (rep1 > maxRep) { saved1 = rep1; rep1 = 0; }
    }
    if (n >= 20) {
        const u32 ilimit = n - 16;
        // immediate rep2 matches after a sequence that ends at ip0 (oracle fast_block :713-722)
        auto rep_loop = [&](u32 ip0) -> u32 {
            if (rep2 > 0) {
                while (ip0 <= ilimit && rd32(src + ip0) == rd32(src + ip0 - rep2)) {
                    const u32 rLength = zcount(src, ip0 + 4, ip0 + 4 - rep2, n) + 4;
                    const u32 t = rep2; rep2 = rep1; rep1 = t;
                    M.put(ip0);
                    ip0 += rLength;
                    M.seqs.push_back({0, 1, rLength});
                }
            }
            return ip0;
        };
        // state of the newest chain batch: a search that starts inside it (behind a sequence of the exact path or the
        // immediate rep2 test) is resumed in the same batch - its lanes below lim are untouched by every table write since
        bool cb = false; u32 pos0 = 0; int lim = 64; u64 okmask = 0, hard = 0, dupu = 0;   // dupu: duplicate lanes nobody has looked at (mode 32)
        u32 h[64], ent[64], eqbv[64]; u64 infov[64]; bool ok[64], hd[64], lg[64];
        const bool resumable = (useChain & 4) != 0;
        // lane j of the batch reads its slot and classifies its candidate (kernel: "probe begin" .. "masks begin")
        auto probe = [&](int j, u32 entry) {
            const u32 p = pos0 + (u32)j;
            ent[j] = entry;
            ok[j] = false; hd[j] = false; lg[j] = false; eqbv[j] = 0; infov[j] = 0;
            if (ent[j] >= prefixIdx) {
                const u32 c = ent[j] - 2;
                if (c - prefixPos >= 4) {
                    u32 e = 0;
                    for (int i = 0; i < 32; ++i) if (src[p - 4 + i] == src[c - 4 + i]) e |= 1u << i;
                    const u32 fwm = ~(e >> 8);
                    const u32 fw = (u32)__builtin_ctz(fwm);                       // <= 24
                    const u32 bkm = ~(e << 28);
                    const u32 bk4 = (u32)__builtin_clz(bkm);                      // <= 4
                    eqbv[j] = e; infov[j] = zchain::pack_info_wide(p - c, fw, zchain::back_code(bk4, c - prefixPos));
                    ok[j] = ((e >> 4) & 15u) == 15u;
                    hd[j] = fw >= 24;                                             // the run leaves the window
                    lg[j] = ok[j] && fw >= 24;
                } else {
                    ok[j] = rd32(src + c) == rd32(src + p);
                    infov[j] = zchain::pack_info_wide(p - c, 0, 0);
                    hd[j] = true;                                                 // candidate within 4 bytes of the prefix start
                }
            }
        };
        auto lane_flags = [&](int j, int lim_) {                                  // okmask / hard bits of a lane that was classified
            okmask &= ~(1ull << j); hard &= ~(1ull << j);
            if (!ok[j]) return;
            okmask |= 1ull << j;
            const int E = j + 4 + (int)(((u32)infov[j] >> 17) & 31u);
            if (hd[j] || (E - 2 >= lim_ && j + 2 >= 64)) hard |= 1ull << j;        // (table writes behind lane 63)
        };
        // table writes of a batch: in lane order, so a slot named by several lanes goes to the latest
        auto commit_table = [&](u64 cm) { for (int j = 0; j < 64; ++j) if ((cm >> j) & 1) M.table[h[j]] = pos0 + (u32)j + 2; };
        for (;;) {
            int ev = 0; u32 e_pos = 0, e_cand = 0, e_sig = 0, e_st = 0, e_half = 0;
            const bool resume = resumable && cb && i0 == 0 && s > pos0 && (int)(s - pos0) + M.resumeMargin <= lim &&
                                (!(useChain & 8) || (okmask >> (s - pos0)) != 0);
            cb = false;
            if (resume || (useChain && i0 == 0 && s >= bs + 4 && rep1 > 0 && (u64)s + 100 < ilimit)) {
                // ---------------- chain batch: 64 consecutive probes ----------------
                const u64 seqs0 = M.stat[1];
                if (resume) M.stat[34]++; else {
                M.stat[0]++;
                pos0 = s;
                for (int j = 0; j < 64; ++j) { h[j] = zhash(src + pos0 + (u32)j, M.hlog, M.mls); probe(j, M.table[h[j]]); }
                u64 dups = 0;
                for (int j = 1; j < 64; ++j) for (int i = 0; i < j; ++i) if (h[i] == h[j]) { dups |= 1ull << j; break; }
                lim = 64; okmask = 0; hard = 0; dupu = 0;
                if (useChain & 32) {
                    // mode 32 (the kernel since round 4): a duplicate lane - its slot is named by an earlier lane of the batch -
                    // does not cut the batch.  The walk gets it as a match that takes the exact path, so it stops there if it
                    // gets there, and the lane is probed again then (below).
                    dupu = dups;
                    for (int j = 0; j < 64; ++j) if (!((dups >> j) & 1)) lane_flags(j, 64);
                    okmask |= dups; hard |= dups;
                } else {
                    if (dups) lim = __builtin_ctzll(dups);
                    for (int j = 0; j < lim; ++j) lane_flags(j, lim);
                }
                }
                u64 eq1 = 0, eq2 = 0;
                for (int j = 0; j < 64; ++j) {
                    const u32 p = pos0 + (u32)j;
                    if (p >= rep1 && src[p] == src[p - rep1]) eq1 |= 1ull << j;
                    if (rep2 && p >= rep2 && src[p] == src[p - rep2]) eq2 |= 1ull << j;
                }
                auto emit = [&](u32 ll, u32 off, u32 ml) {
                    M.seqs.push_back({ll, off, ml}); M.litSize += ll; M.stat[1]++;
                };
                zchain::WalkOut w = zchain::walk(okmask, hard, lim, pos0, (int32_t)anchor - (int32_t)pos0, rep1, rep2, eq1, eq2,
                    [&](int j) { return infov[j]; }, [&](int j) { return eqbv[j]; }, emit, resume ? (int)(s - pos0) : 0, 0);
                for (;;) {
                    if (w.outcome == zchain::ZC_SPECIAL && w.ev == 1 && ((dupu >> w.el) & 1)) {
                        // the search has reached a duplicate lane: every write the reference made before it reads this slot is
                        // known now - they go to the table, the lane probes again, the walk starts over at the search's first lane
                        M.stat[37]++;
                        const int el = w.el;
                        // (the probes of the search in progress, lanes [S, el), are NOT written yet: the walk may still ask for a
                        //  fresh batch at S, which probes them again - the latest of them on the slot is the candidate, from registers)
                        commit_table(w.commit & ((1ull << w.S) - 1));
                        int il = -1;
                        for (int y = w.S; y < el; ++y) if (h[y] == h[el]) il = y;
                        probe(el, il >= 0 ? pos0 + (u32)il + 2 : M.table[h[el]]);
                        lane_flags(el, 64); dupu &= ~(1ull << el);
                        w.outcome = zchain::ZC_REFRESH; w.commit &= (1ull << w.S) - 1;
                    }
                    if (w.outcome != zchain::ZC_REFRESH) break;
                    M.stat[33]++;
                    u64 e1 = 0, e2 = 0;
                    for (int j = 0; j < 64; ++j) {
                        const u32 p = pos0 + (u32)j;
                        if (p >= w.rep1 && src[p] == src[p - w.rep1]) e1 |= 1ull << j;           // (lanes below S may lie before the offset's reach)
                        if (w.rep2 && p >= w.rep2 && src[p] == src[p - w.rep2]) e2 |= 1ull << j;
                    }
                    w = zchain::walk(okmask, hard, lim, pos0, w.arel, w.rep1, w.rep2, e1, e2,
                        [&](int j) { return infov[j]; }, [&](int j) { return eqbv[j]; }, emit, w.S, w.commit);
                }
                if (resume && M.stat[1] > seqs0) M.stat[35]++;
                commit_table(w.commit);
                if (g_trace && g_trace_n < 4096) {
                    u64* t = g_trace + 8 * g_trace_n++;
                    t[0] = pos0; t[1] = (u32)lim; t[2] = okmask; t[3] = w.commit; t[4] = (u32)w.outcome | ((u64)(u32)w.S << 8) | ((u64)(u32)w.ev << 16) | ((u64)(u32)w.el << 24) | ((u64)(u32)w.ntail << 32) | ((u64)w.post << 40);
                    t[5] = eq1; t[6] = eq2; t[7] = ((u64)rep1 << 32) | rep2;
                }
                anchor = (u32)((int32_t)pos0 + w.arel); rep1 = w.rep1; rep2 = w.rep2; s = pos0 + (u32)w.S;
                if (w.outcome == zchain::ZC_RESTART) {
                    M.stat[3]++;
                    for (int t = 3 - w.ntail; t < 3; ++t) M.put(pos0 + (u32)w.tail[t]);
                    if (w.ntail) M.stat[6]++;
                    if (w.post) { M.stat[7]++; s = rep_loop(s); anchor = s; cb = true; }
                    continue;
                }
                if (w.outcome == zchain::ZC_CONTINUE) { M.stat[4]++; i0 = (u32)(w.R - w.S); }
                else if ((useChain & 16) && w.ev == 1 && lg[w.el] && w.el <= 59 &&
                         ((((u32)infov[w.el] >> 22) & 7u) < 5u || (u32)(w.el - w.arel) < 5u) &&
                         [&]() {
                    // a match that runs out of its lane's 28-byte window but ends inside the batch: its length is the run of the
                    // equality map at the match's own distance (one byte load per lane) - no exact path, and the parse goes on in
                    // this batch (the search behind it is resumed)
                    const int j = w.el;
                    const u32 off = (u32)(infov[j] >> 32), bk = ((u32)infov[j] >> 22) & 7u;
                    const u32 room = (u32)(j - w.arel), back = bk < room ? bk : room;
                    u64 m = 0;
                    for (int i = 0; i < 64; ++i) { const u32 p = pos0 + (u32)i; if (p >= off && src[p] == src[p - off]) m |= 1ull << i; }
                    const u32 run = zchain::ctz64u(~(m >> (j + 4))), E = (u32)j + 4 + run;
                    if (E > 63) return false;                                             // reaches the end of the batch: length unknown
                    M.stat[36]++;
                    const u32 ip0 = pos0 + (u32)j - back;
                    M.seqs.push_back({ip0 - anchor, off + 3, 4 + back + run}); M.litSize += ip0 - anchor;
                    rep2 = rep1; rep1 = off;
                    M.put(pos0 + (u32)j + 1); M.put(pos0 + (u32)j + 2); M.put(pos0 + E - 2);
                    s = rep_loop(pos0 + E); anchor = s;
                    return true;
                }()) { cb = true; continue; }
                else {
                    M.stat[2]++; M.stat[8 + (w.why & 31)]++; cb = true;
                    ev = w.ev; e_pos = pos0 + (u32)w.el; e_cand = ent[w.el]; e_sig = 2; e_st = 2; e_half = (u32)(w.el - w.S) & 1u;
                }
            }
            if (!ev) {
                // ---------------- sequential search from probe i0 of the search that starts at s ----------------
                M.stat[5]++;
                for (u32 K = i0;; ++K) {
                    const u32 m = K >> 1, half = K & 1;
                    const u32 am = g_sched.a[m], sg = g_sched.sigma[m], st = g_sched.st[m];
                    if (!((u64)s + am + sg + 1 < ilimit)) break;
                    const u32 pos = s + am + half;
                    const u32 hh = zhash(src + pos, M.hlog, M.mls);
                    const u32 idx = M.table[hh];
                    M.table[hh] = pos + 2;
                    if (half == 0 && rep1 > 0 && rd32(src + pos + sg) == rd32(src + pos + sg - rep1)) { ev = 2; e_pos = pos; e_sig = sg; e_st = st; e_half = 0; break; }
                    if (idx >= prefixIdx && rd32(src + idx - 2) == rd32(src + pos)) { ev = 1; e_pos = pos; e_cand = idx; e_sig = sg; e_st = st; e_half = half; break; }
                }
                i0 = 0;
                if (!ev) break;
            }
            // ---------------- one sequence the exact way (oracle fast_block :701-723) ----------------
            {
                const u32 ip1 = e_half == 0 ? e_pos + 1 : e_pos - 1 + e_sig;
                if (e_half == 0 || e_st <= 4) M.put(ip1);
            }
            const u32 current0 = e_pos;
            u32 ip0, match0, offcode, mLength;
            if (ev == 2) {
                ip0 = e_pos + e_sig; match0 = ip0 - rep1;
                const u32 back = src[ip0 - 1] == src[match0 - 1];
                ip0 -= back; match0 -= back; offcode = 1; mLength = back + 4;
            } else {
                ip0 = e_pos; match0 = e_cand - 2;
                rep2 = rep1; rep1 = ip0 - match0; offcode = rep1 + 3; mLength = 4;
                while (ip0 > anchor && match0 > prefixPos && src[ip0 - 1] == src[match0 - 1]) { ip0--; match0--; mLength++; }
            }
            mLength += zcount(src, ip0 + mLength, match0 + mLength, n);
            M.seqs.push_back({ip0 - anchor, offcode, mLength}); M.litSize += ip0 - anchor;
            ip0 += mLength; anchor = ip0;
            if (ip0 <= ilimit) {
                M.put(current0 + 2);
                M.put(ip0 - 2);
                ip0 = rep_loop(ip0); anchor = ip0;
            }
            s = ip0;
        }
    }
    saved2 = (saved1 != 0 && rep1 != 0) ? saved1 : saved2;
    rep1io = rep1 ? rep1 : saved1;
    rep2io = rep2 ? rep2 : saved2;
    M.litSize += n - anchor;
    M.lastLits = n - anchor;
}

extern "C" size_t zchain_model_sequences(const u8* src, size_t n_, u32* out, size_t* litSizeOut, u64* stats, int useChain)
{
    Model M; M.src = src; M.n = (u32)n_;
    if (useChain >> 8) M.resumeMargin = useChain >> 8;
    useChain &= 255;
    const u32 n = M.n;
    {   // level1_params (oracle/zstd_oracle.c)
        const u32 srcLog = (n < 64) ? 6 : (u32)hb(n - 1) + 1;
        if (n <= 16384) { const u32 w = srcLog < 14 ? srcLog : 14; M.hlog = (15 < w + 1) ? 15 : w + 1; M.mls = 5; }
        else { const u32 w = srcLog < 17 ? srcLog : 17; M.hlog = (13 < w + 1) ? 13 : w + 1; M.mls = 6; }
    }
    M.table.assign((size_t)1 << M.hlog, 0);
    {   const Win win = { 2, 2, 1u << 31 };
        u32 r1 = 1, r2 = 4;
        model_block(M, 0, n, win, r1, r2, useChain);
    }
    for (size_t i = 0; i < M.seqs.size(); ++i) { out[4 * i] = M.seqs[i].ll; out[4 * i + 1] = M.seqs[i].off; out[4 * i + 2] = M.seqs[i].ml; out[4 * i + 3] = 0; }
    *litSizeOut = (size_t)M.litSize;
    if (stats) memcpy(stats, M.stat, 8 * sizeof(u64));
    if (stats && (useChain & 2)) memcpy(stats, M.stat, sizeof(M.stat));
    return M.seqs.size();
}

// all blocks of a frame (fast strategy): rows {ll, off, ml, 0}, and {0xFFFFFFFF, block, last literals, 0} after every block.
// Every block counts as confirmed (repeat offsets carried on), as in orc_dbg_zstd_frame_sequences.
extern "C" size_t zchain_model_frame_sequences(const u8* src, size_t n_, u32 windowLog, u32 hashLog, u32 mls, u32* out, int useChain)
{
    Model M; M.src = src; M.n = (u32)n_; M.hlog = hashLog; M.mls = mls;
    M.table.assign((size_t)1 << hashLog, 0);
    const u32 n = M.n, blockMax = windowLog < 17 ? 1u << windowLog : 128u << 10;
    Win win = { 2, 2, 1u << windowLog };
    u32 r1 = 1, r2 = 4; size_t rows = 0; u32 blk = 0;
    for (u32 bs = 0; bs < n; bs += blockMax, ++blk) {
        const u32 bn = n - bs < blockMax ? n - bs : blockMax;
        if (bs + 2 > win.maxDist) {
            const u32 low = bs + 2 - win.maxDist;
            if (win.lowLimit < low) win.lowLimit = low;
            if (win.dictLimit < win.lowLimit) win.dictLimit = win.lowLimit;
        }
        M.seqs.clear();
        if (bn >= 7) model_block(M, bs, bs + bn, win, r1, r2, useChain); else M.lastLits = bn;
        for (const Seq& q : M.seqs) { out[4 * rows] = q.ll; out[4 * rows + 1] = q.off; out[4 * rows + 2] = q.ml; out[4 * rows + 3] = 0; rows++; }
        out[4 * rows] = 0xFFFFFFFFu; out[4 * rows + 1] = blk; out[4 * rows + 2] = M.lastLits; out[4 * rows + 3] = 0; rows++;
    }
    return rows;
}

// zero_bytes8 (the multiply-gather form the kernel uses) against the plain per-dword zero_bytes on n pseudo-random pairs with
// zero bytes planted at random; returns the number of disagreements
extern "C" size_t zchain_model_zero_bytes8_check(size_t n, u64 seed)
{
    size_t bad = 0;
    u64 x = seed | 1;
    auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (size_t i = 0; i < n; ++i) {
        const u64 r = next(), m = next();
        u32 a = (u32)r, b = (u32)(r >> 32);
        for (int k = 0; k < 4; ++k) { if ((m >> k) & 1) a &= ~(0xFFu << (8 * k)); if ((m >> (8 + k)) & 1) b &= ~(0xFFu << (8 * k)); }
        if ((zchain::zero_bytes(a) | (zchain::zero_bytes(b) << 4)) != zchain::zero_bytes8(a, b)) bad++;
    }
    return bad;
}
