// csrc/rap.hip -- RAP ("random access point") streams: the reference's multi-threaded container, written from and
// decoded into GPU chunk batches.
//
// Reference: threads/threads.h:20-112 (frame layout), threads/threads.c:46-153,183-300 (setup / partition metadata),
// lz4.c:2655-2909 (AOCL_LZ4_compress_fast_mt: RAP header + last-literals join), snappy.cc:2505-2654 (varint join),
// zstd_compress.c:5343-5385,5499-5517 (skippable frame + frame concatenation), and the MT decoders
// lz4.c:4785-4890, snappy.cc:2280-2390, zstd_decompress.c.
//
// A partition = one GPU chunk.  The per-partition compressors of the reference's MT path make exactly the parse of
// the one-shot functions (the LZ4 one only leaves the final literal run out, lz4.c:2293-2337), so the partitions'
// payloads are cut out of the chunk slots; the join of the reference (serial memcpy + first-token rewrite) becomes a
// plan of three copies per partition (rewritten header, carried literals from the source, payload body) executed by
// one wavefront per partition.  The plan itself (12 bytes of metadata per partition) is made on the host.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <chrono>
#include <mutex>
#include <vector>
#include <stdio.h>
#include <stdlib.h>
#include "common.hip.h"
#include "host_util.h"

namespace aoclgpu {
namespace {

constexpr uint64_t kRapMagic = 0x434C4C5F4C434F41ull;       // "AOCL_LLC" (threads/threads.h:79)
constexpr uint32_t kRapHead = 8 + 4 + 2 + 2;                // magic, frame length, main / child thread counts
constexpr uint32_t kRapEntry = 12;                          // offset, length, decoded length
constexpr uint32_t kZstdSkipMagic = 0x184D2A50u;            // ZSTD_MAGIC_SKIPPABLE_START, variant 0 (zstd_compress.c:5353)
constexpr uint32_t kHead = 16;                              // bytes of every slot fetched for the first-token parse

struct CopyOp {                                             // one contiguous piece of the output stream
    uint64_t dst;                                           // offset in the stream
    uint64_t src;                                           // offset in the header blob (from == 0) or in the slot buffer (from == 1)
    uint32_t len, from;
};

__global__ void __launch_bounds__(64) rap_gather_kernel(const CopyOp* __restrict__ ops, uint32_t n,
                                                        const uint8_t* __restrict__ blob, const uint8_t* __restrict__ slots,
                                                        uint8_t* __restrict__ out)
{
    const uint32_t k = blockIdx.x;
    if (k >= n) return;
    const CopyOp p = ops[k];
    wave_copy((gbyte*)out + p.dst, (const gbyte*)(p.from ? slots : blob) + p.src, p.len, lane_id());
}

__global__ void rap_heads_kernel(const uint8_t* __restrict__ slots, const uint64_t* __restrict__ slot_off, const int32_t* __restrict__ len,
                                 uint32_t n, uint8_t* __restrict__ heads)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint8_t* s = slots + slot_off[k];
    const uint32_t l = len[k] > 0 ? (uint32_t)len[k] : 0;
    for (uint32_t i = 0; i < kHead; ++i) heads[(uint64_t)k * kHead + i] = i < l ? s[i] : 0;
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    bool alloc(size_t n) { return hipMalloc(&p, n ? n : 1) == hipSuccess; }
};

uint32_t varint_len(uint32_t v) { uint32_t n = 1; while (v >= 128) { v >>= 7; n++; } return n; }
void put_varint(std::vector<uint8_t>& o, uint32_t v) { while (v >= 128) { o.push_back((uint8_t)(v | 128)); v >>= 7; } o.push_back((uint8_t)v); }
void put32(uint8_t* p, uint32_t v) { memcpy(p, &v, 4); }

#define HIPCK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_error(#call, e_); return AOCLGPU_ERR_HIP; } } while (0)

}  // namespace
}  // namespace aoclgpu

using namespace aoclgpu;

extern "C" uint64_t aoclgpu_rap_frame_bytes(aoclgpu_codec codec, uint32_t nchunks)
{
    if (nchunks <= 1) return 0;
    return (uint64_t)kRapHead + (uint64_t)kRapEntry * nchunks + (codec == AOCLGPU_ZSTD ? 8 : 0);
}

namespace {
// The join of the reference's MT compressors for K partitions whose one-shot streams lie in `slots` (device) at
// slotoff[k], olen[k] bytes each; n[k] = uncompressed size, aux[k] = length of the partition's final literal run (LZ4).
// Everything the join needs is IN the compressed partitions: the literals LZ4 carries from partition k into the first
// sequence of partition k+1 are the last aux[k] bytes of partition k's stream (its final sequence is token, length
// bytes, literals), so the partitions may have been compressed on another GPU and only their streams gathered here.
int rap_join(aoclgpu_codec codec, uint32_t K, const int32_t* olen, const uint32_t* n, const uint32_t* aux, const uint64_t* slotoff,
             const uint8_t* slots, uint8_t* out, uint64_t cap, uint64_t* stream_bytes, hipStream_t s)
{
    uint64_t total_src = 0;
    for (uint32_t k = 0; k < K; ++k) {
        if (olen[k] <= 0) { set_error_msg("rap: a chunk failed to compress"); return AOCLGPU_ERR_ARG; }
        total_src += n[k];
    }
    if (total_src > 0x7FFFFFFFull) { set_error_msg("rap: RAP offsets are 32-bit, at most 2 GiB per stream"); return AOCLGPU_ERR_ARG; }

    std::vector<CopyOp> ops;
    ops.reserve((size_t)K * 3);
    std::vector<uint8_t> blob;                                   // frame + rewritten headers
    const uint32_t frame = (uint32_t)aoclgpu_rap_frame_bytes(codec, K);
    blob.resize(frame, 0);
    uint64_t pos = frame;                                        // running output offset
    const uint32_t base = (codec == AOCLGPU_ZSTD && K > 1) ? 8u : 0u;   // RAP offsets count from the magic word
    auto entry = [&](uint32_t k, uint64_t off, uint32_t len, uint32_t dlen) {
        uint8_t* e = blob.data() + base + kRapHead + (size_t)kRapEntry * k;
        put32(e, (uint32_t)(off - base)); put32(e + 4, len); put32(e + 8, dlen);
    };
    auto emit = [&](uint64_t src, uint32_t len, uint32_t from) { if (len) { ops.push_back(CopyOp{pos, src, len, from}); pos += len; } };
    if (K == 1) {                                                // one partition: the plain one-shot stream
        pos = 0;
        emit(slotoff[0], (uint32_t)olen[0], 1);
    } else if (codec == AOCLGPU_ZSTD) {
        for (uint32_t k = 0; k < K; ++k) {
            entry(k, pos, (uint32_t)olen[k], n[k]);
            emit(slotoff[k], (uint32_t)olen[k], 1);
        }
    } else if (codec == AOCLGPU_SNAPPY) {
        const uint32_t hoff = (uint32_t)blob.size();
        put_varint(blob, (uint32_t)total_src);                   // one length prefix for the whole stream
        emit(hoff, (uint32_t)blob.size() - hoff, 0);
        for (uint32_t k = 0; k < K; ++k) {
            const uint32_t w = varint_len(n[k]);
            if ((uint32_t)olen[k] < w) { set_error_msg("rap: snappy chunk shorter than its prefix"); return AOCLGPU_ERR_ARG; }
            const uint32_t body = (uint32_t)olen[k] - w;
            entry(k, pos, body, n[k]);
            emit(slotoff[k] + w, body, 1);
        }
    } else {                                                     // LZ4 (lz4.c:2736-2906)
        std::vector<uint8_t> heads((size_t)K * kHead);
        {
            DevBuf dh, doff, dlen;
            if (!dh.alloc((size_t)K * kHead) || !doff.alloc((size_t)K * 8) || !dlen.alloc((size_t)K * 4)) { set_error_msg("rap: hipMalloc"); return AOCLGPU_ERR_HIP; }
            HIPCK(hipMemcpyAsync(doff.p, slotoff, (size_t)K * 8, hipMemcpyHostToDevice, s));
            HIPCK(hipMemcpyAsync(dlen.p, olen, (size_t)K * 4, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(rap_heads_kernel, dim3((K + 255) / 256), dim3(256), 0, s, slots, (const uint64_t*)doff.p, (const int32_t*)dlen.p, K, (uint8_t*)dh.p);
            if (int rc = check_launch("rap_heads_kernel")) return rc;
            HIPCK(hipStreamSynchronize(s));
            HIPCK(hipMemcpy(heads.data(), dh.p, heads.size(), hipMemcpyDeviceToHost));
        }
        struct Seg { uint64_t src; uint32_t len; };
        std::vector<Seg> carry;                                  // literal runs waiting for the next partition's first sequence
        uint64_t pend = 0;                                       // their total length
        uint32_t prev_len = 0; uint64_t prev_off = frame;
        for (uint32_t k = 0; k < K; ++k) {
            const bool last = k + 1 == K;
            const uint32_t L = last ? 0 : aux[k];
            const uint32_t lastseq = last ? 0 : 1 + (L < 15 ? 0 : 1 + (L - 15) / 255) + L;
            if ((uint32_t)olen[k] < lastseq || L > n[k]) { set_error_msg("rap: inconsistent aux[] (last literal run)"); return AOCLGPU_ERR_ARG; }
            const uint32_t body_end = (uint32_t)olen[k] - lastseq;      // dst_trap_size of the reference's MT compressor
            const Seg tail = {slotoff[k] + (uint32_t)olen[k] - L, L};   // the final literal run's bytes
            const uint64_t off = prev_off + prev_len;
            if (k == 0) {
                entry(0, pos, body_end, n[0] - L);
                prev_off = pos; prev_len = body_end;
                emit(slotoff[0], body_end, 1);
                carry.assign(1, tail); pend = L;
                continue;
            }
            if (!last && body_end == 0) {                        // all literals: they travel on to the next partition (:2808-2823)
                entry(k, off, 0, 0);
                prev_off = off; prev_len = 0;
                carry.push_back(tail); pend += L;
                continue;
            }
            // rewrite the first token: its literal run grows by the carried literals (:2826-2868)
            const uint8_t* h = heads.data() + (size_t)k * kHead;
            std::vector<uint8_t> big;
            const uint32_t tok = h[0];
            uint64_t cur_lit = tok >> 4; uint32_t hdr_in = 1;
            if (cur_lit == 15) {
                const uint8_t* q = h; uint32_t avail = kHead;
                for (;;) {
                    if (hdr_in >= avail) {                       // a very long first literal run: fetch more of the slot
                        if (!big.empty() || (uint32_t)olen[k] <= kHead) { set_error_msg("rap: truncated first token"); return AOCLGPU_ERR_ARG; }
                        big.resize((uint32_t)olen[k] < 4096u ? (uint32_t)olen[k] : 4096u);
                        HIPCK(hipMemcpy(big.data(), slots + slotoff[k], big.size(), hipMemcpyDeviceToHost));
                        q = big.data(); avail = (uint32_t)big.size();
                    }
                    const uint32_t e = q[hdr_in++]; cur_lit += e;
                    if (e != 255) break;
                }
            }
            const uint64_t new_lit = cur_lit + pend;
            const uint32_t hoff = (uint32_t)blob.size();
            blob.push_back((uint8_t)(((new_lit < 15 ? new_lit : 15) << 4) | (tok & 15)));
            if (new_lit >= 15) { uint64_t a = new_lit - 15; for (; a >= 255; a -= 255) blob.push_back(255); blob.push_back((uint8_t)a); }
            const uint32_t hlen = (uint32_t)blob.size() - hoff;
            if (body_end < hdr_in) { set_error_msg("rap: first token longer than the partition"); return AOCLGPU_ERR_ARG; }
            const uint32_t body = body_end - hdr_in;
            const uint32_t plen = hlen + (uint32_t)pend + body;
            entry(k, off, plen, (uint32_t)((n[k] - L) + pend));
            emit(hoff, hlen, 0);
            for (const Seg& g : carry) emit(g.src, g.len, 1);
            emit(slotoff[k] + hdr_in, body, 1);
            prev_off = off; prev_len = plen;
            carry.assign(1, tail); pend = L;
        }
    }
    if (K > 1) {                                                 // frame header (threads/threads.c:104-111)
        uint8_t* f = blob.data();
        if (codec == AOCLGPU_ZSTD) { put32(f, kZstdSkipMagic); put32(f + 4, frame - 8); f += 8; }
        memcpy(f, &kRapMagic, 8);
        put32(f + 8, frame - base);
        put32(f + 12, K);                                        // u16 partitions, u16 child threads = 0
    }
    if (pos > cap) { set_error_msg("rap: stream_cap too small"); return AOCLGPU_ERR_ARG; }
    if (pos > 0xFFFFFFFFull) { set_error_msg("rap: stream exceeds 32-bit RAP offsets"); return AOCLGPU_ERR_ARG; }
    // ---- execute: the frame, then one wavefront per piece ----
    DevBuf dp, db;
    if (!dp.alloc(ops.size() * sizeof(CopyOp)) || !db.alloc(blob.size())) { set_error_msg("rap: hipMalloc"); return AOCLGPU_ERR_HIP; }
    if (!ops.empty()) HIPCK(hipMemcpyAsync(dp.p, ops.data(), ops.size() * sizeof(CopyOp), hipMemcpyHostToDevice, s));
    if (!blob.empty()) HIPCK(hipMemcpyAsync(db.p, blob.data(), blob.size(), hipMemcpyHostToDevice, s));
    if (frame) HIPCK(hipMemcpyAsync(out, db.p, frame, hipMemcpyDeviceToDevice, s));
    if (!ops.empty()) {
        hipLaunchKernelGGL(rap_gather_kernel, dim3((uint32_t)ops.size()), dim3(64), 0, s, (const CopyOp*)dp.p, (uint32_t)ops.size(),
                           (const uint8_t*)db.p, slots, out);
        if (int rc = check_launch("rap_gather_kernel")) return rc;
    }
    HIPCK(hipStreamSynchronize(s));                              // the plan buffers die with this call
    *stream_bytes = pos;
    return AOCLGPU_OK;
}
}  // namespace

extern "C" int aoclgpu_rap_pack(aoclgpu_codec codec, const aoclgpu_batch* b, uint8_t* out, uint64_t cap,
                                uint64_t* stream_bytes, void* stream)
{
    if (int rc = check_batch(b, false)) return rc;
    if (!out || !stream_bytes || b->nchunks == 0) { set_error_msg("rap_pack: NULL output / empty batch"); return AOCLGPU_ERR_ARG; }
    if (b->nchunks > 65535) { set_error_msg("rap_pack: more than 65535 partitions (16-bit count, threads/threads.h:82)"); return AOCLGPU_ERR_ARG; }
    if (codec == AOCLGPU_LZ4 && b->nchunks > 1 && !b->aux) { set_error_msg("rap_pack: LZ4 needs the aux[] output of the compress launch"); return AOCLGPU_ERR_ARG; }
    const uint32_t K = b->nchunks;
    hipStream_t s = (hipStream_t)stream;
    HIPCK(hipStreamSynchronize(s));
    // ---- per-chunk geometry and results, on the host ----
    std::vector<int32_t> olen(K);
    std::vector<uint32_t> aux(K, 0), n(K);
    std::vector<uint64_t> srcoff(K), slotoff(K);
    HIPCK(hipMemcpy(olen.data(), b->out_len, (size_t)K * 4, hipMemcpyDeviceToHost));
    if (codec == AOCLGPU_LZ4 && K > 1) HIPCK(hipMemcpy(aux.data(), b->aux, (size_t)K * 4, hipMemcpyDeviceToHost));
    if (b->src_off) {
        HIPCK(hipMemcpy(srcoff.data(), b->src_off, (size_t)K * 8, hipMemcpyDeviceToHost));
        HIPCK(hipMemcpy(slotoff.data(), b->dst_off, (size_t)K * 8, hipMemcpyDeviceToHost));
        HIPCK(hipMemcpy(n.data(), b->src_len, (size_t)K * 4, hipMemcpyDeviceToHost));
        for (uint32_t k = 1; k < K; ++k)
            if (srcoff[k] != srcoff[k - 1] + n[k - 1]) { set_error_msg("rap_pack: partitions must be consecutive in the source"); return AOCLGPU_ERR_ARG; }
    } else {
        for (uint32_t k = 0; k < K; ++k) {
            srcoff[k] = (uint64_t)k * b->chunk_bytes; slotoff[k] = (uint64_t)k * b->slot_bytes;
            const uint64_t rem = b->total_bytes > srcoff[k] ? b->total_bytes - srcoff[k] : 0;
            n[k] = (uint32_t)(rem < b->chunk_bytes ? rem : b->chunk_bytes);
        }
    }
    return rap_join(codec, K, olen.data(), n.data(), aux.data(), slotoff.data(), b->dst, out, cap, stream_bytes, s);
}

extern "C" int aoclgpu_rap_assemble(aoclgpu_codec codec, const uint8_t* gathered, const uint64_t* chunk_off, const int32_t* chunk_len,
                                    const uint32_t* chunk_src_len, const uint32_t* chunk_aux, uint32_t nchunks,
                                    uint8_t* out, uint64_t cap, uint64_t* stream_bytes, void* stream)
{
    if (!gathered || !chunk_off || !chunk_len || !chunk_src_len || !out || !stream_bytes || nchunks == 0) { set_error_msg("rap_assemble: NULL argument / no chunks"); return AOCLGPU_ERR_ARG; }
    if (nchunks > 65535) { set_error_msg("rap_assemble: more than 65535 partitions (16-bit count, threads/threads.h:82)"); return AOCLGPU_ERR_ARG; }
    if (codec == AOCLGPU_LZ4 && nchunks > 1 && !chunk_aux) { set_error_msg("rap_assemble: LZ4 needs the gathered aux[] (final literal run lengths)"); return AOCLGPU_ERR_ARG; }
    std::vector<uint32_t> zero;
    if (!chunk_aux) { zero.assign(nchunks, 0); chunk_aux = zero.data(); }
    hipStream_t s = (hipStream_t)stream;
    HIPCK(hipStreamSynchronize(s));
    return rap_join(codec, nchunks, chunk_len, chunk_src_len, chunk_aux, chunk_off, gathered, out, cap, stream_bytes, s);
}

// The whole sharded stream as the SEQUENCE of RAP streams the formats allow: a RAP frame holds 32-bit offsets and a 16-bit
// partition count (threads/threads.h:77-106), and the reference's own bench feeds its codecs slabs of at most 1 GiB
// (MAX_MEM_SIZE_FOR_FILE_READ, test/codec_bench.h:58).  Consecutive chunks are grouped into slabs of at most `slab_src_bytes`
// source bytes (0 = 1 GiB; at most 2 GiB - 1) and 65535 partitions; every slab is one aoclgpu_rap_assemble stream, the
// streams are written back to back into `out`.  slab_off[0 .. *nslabs] (host, room for max_slabs + 1 entries) = their
// offsets in `out`; slab_first[0 .. *nslabs] = the first chunk of each slab.  A decoder walks the slabs with
// aoclgpu_rap_unpack / the reference's MT decoders, one call per slab.
extern "C" int aoclgpu_rap_assemble_slabs(aoclgpu_codec codec, const uint8_t* gathered, const uint64_t* chunk_off, const int32_t* chunk_len,
                                          const uint32_t* chunk_src_len, const uint32_t* chunk_aux, uint64_t nchunks, uint64_t slab_src_bytes,
                                          uint8_t* out, uint64_t cap, uint64_t* slab_off, uint64_t* slab_first, uint32_t max_slabs, uint32_t* nslabs,
                                          void* stream)
{
    if (!gathered || !chunk_off || !chunk_len || !chunk_src_len || !out || !slab_off || !slab_first || !nslabs || nchunks == 0) {
        set_error_msg("rap_assemble_slabs: NULL argument / no chunks"); return AOCLGPU_ERR_ARG;
    }
    if (slab_src_bytes == 0) slab_src_bytes = 1ull << 30;
    if (slab_src_bytes > 0x7FFFFFFFull) { set_error_msg("rap_assemble_slabs: a slab holds at most 2 GiB - 1 of source (32-bit RAP fields)"); return AOCLGPU_ERR_ARG; }
    uint64_t pos = 0, first = 0; uint32_t k = 0;
    while (first < nchunks) {
        uint64_t last = first, bytes = 0;
        while (last < nchunks && last - first < 65535 && bytes + chunk_src_len[last] <= slab_src_bytes) { bytes += chunk_src_len[last]; ++last; }
        if (last == first) { set_error_msg("rap_assemble_slabs: a chunk is larger than a slab"); return AOCLGPU_ERR_ARG; }
        if (k >= max_slabs) { set_error_msg("rap_assemble_slabs: more slabs than max_slabs"); return AOCLGPU_ERR_ARG; }
        uint64_t sb = 0;
        const int rc = aoclgpu_rap_assemble(codec, gathered, chunk_off + first, chunk_len + first, chunk_src_len + first,
                                            chunk_aux ? chunk_aux + first : nullptr, (uint32_t)(last - first), out + pos, cap - pos, &sb, stream);
        if (rc) return rc;
        slab_off[k] = pos; slab_first[k] = first;
        pos += sb; first = last; ++k;
    }
    slab_off[k] = pos; slab_first[k] = nchunks;
    *nslabs = k;
    return AOCLGPU_OK;
}

extern "C" int aoclgpu_lz4_decompress_chunks(const aoclgpu_batch* b, void* stream);
extern "C" int aoclgpu_snappy_decompress_chunks(const aoclgpu_batch* b, void* stream);
extern "C" int aoclgpu_zstd_decompress_chunks(const aoclgpu_batch* b, void* stream);

extern "C" int aoclgpu_rap_unpack(aoclgpu_codec codec, const uint8_t* in, uint64_t nbytes, uint8_t* dst, uint64_t dst_cap,
                                  int64_t* decoded, void* stream)
{
    if (!in || !dst || !decoded) { set_error_msg("rap_unpack: NULL argument"); return AOCLGPU_ERR_ARG; }
    *decoded = -1;
    hipStream_t s = (hipStream_t)stream;
    HIPCK(hipStreamSynchronize(s));
    uint8_t head[24] = {0};
    HIPCK(hipMemcpy(head, in, nbytes < sizeof(head) ? nbytes : sizeof(head), hipMemcpyDeviceToHost));
    uint32_t base = 0;
    if (codec == AOCLGPU_ZSTD && nbytes >= 24) {                 // the RAP frame travels in a skippable frame
        uint32_t m; memcpy(&m, head, 4);
        if (m == kZstdSkipMagic) base = 8;
    }
    uint64_t magic = 0;
    if (nbytes >= base + kRapHead) memcpy(&magic, head + base, 8);
    std::vector<uint64_t> soff, doff; std::vector<uint32_t> slen, dcap;
    uint64_t total = 0;
    bool rap = magic == kRapMagic;
    if (!rap) {                                                  // threads/threads.c:194-203: not a RAP stream -> one partition
        if (nbytes > 0xFFFFFFFFull) { set_error_msg("rap_unpack: plain stream too large"); return AOCLGPU_ERR_ARG; }
        soff.push_back(0); slen.push_back((uint32_t)nbytes); doff.push_back(0);
        dcap.push_back((uint32_t)(dst_cap > 0x7FFFFFFFull ? 0x7FFFFFFFull : dst_cap));
    } else {
        uint32_t flen, cnt; memcpy(&flen, head + base + 8, 4); memcpy(&cnt, head + base + 12, 4);
        const uint32_t K = cnt & 0xFFFFu;
        if (K == 0 || (uint64_t)base + kRapHead + (uint64_t)kRapEntry * K > nbytes || flen < kRapHead + kRapEntry * K) return AOCLGPU_OK;   // *decoded = -1
        std::vector<uint8_t> ent((size_t)kRapEntry * K);
        HIPCK(hipMemcpy(ent.data(), in + base + kRapHead, ent.size(), hipMemcpyDeviceToHost));
        for (uint32_t k = 0; k < K; ++k) {
            uint32_t off, len, dl; memcpy(&off, &ent[12 * k], 4); memcpy(&len, &ent[12 * k + 4], 4); memcpy(&dl, &ent[12 * k + 8], 4);
            if ((uint64_t)base + off + len > nbytes || total + dl > dst_cap || dl > 0x7FFFFFFFu) return AOCLGPU_OK;
            if (len == 0) { if (dl != 0) return AOCLGPU_OK; continue; }           // empty partition (lz4.c:2808-2823)
            soff.push_back((uint64_t)base + off); slen.push_back(len); doff.push_back(total); dcap.push_back(dl);
            total += dl;
        }
        if (soff.empty()) { *decoded = 0; return AOCLGPU_OK; }
    }
    const uint32_t K = (uint32_t)soff.size();
    DevBuf a, bq, c, d, o;
    if (!a.alloc((size_t)K * 8) || !bq.alloc((size_t)K * 4) || !c.alloc((size_t)K * 8) || !d.alloc((size_t)K * 4) || !o.alloc((size_t)K * 4)) {
        set_error_msg("rap_unpack: hipMalloc"); return AOCLGPU_ERR_HIP;
    }
    HIPCK(hipMemcpyAsync(a.p, soff.data(), (size_t)K * 8, hipMemcpyHostToDevice, s));
    HIPCK(hipMemcpyAsync(bq.p, slen.data(), (size_t)K * 4, hipMemcpyHostToDevice, s));
    HIPCK(hipMemcpyAsync(c.p, doff.data(), (size_t)K * 8, hipMemcpyHostToDevice, s));
    HIPCK(hipMemcpyAsync(d.p, dcap.data(), (size_t)K * 4, hipMemcpyHostToDevice, s));
    aoclgpu_batch bt; memset(&bt, 0, sizeof(bt));
    bt.src = in; bt.dst = dst; bt.src_off = (const uint64_t*)a.p; bt.src_len = (const uint32_t*)bq.p;
    bt.dst_off = (const uint64_t*)c.p; bt.dst_cap = (const uint32_t*)d.p; bt.out_len = (int32_t*)o.p; bt.nchunks = K;
    bt.reserved = rap ? AOCLGPU_FLAG_RAP_PARTITIONS : 0u;
    int rc = codec == AOCLGPU_LZ4 ? aoclgpu_lz4_decompress_chunks(&bt, stream)
           : codec == AOCLGPU_SNAPPY ? aoclgpu_snappy_decompress_chunks(&bt, stream) : aoclgpu_zstd_decompress_chunks(&bt, stream);
    if (rc) return rc;
    HIPCK(hipStreamSynchronize(s));
    std::vector<int32_t> res(K);
    HIPCK(hipMemcpy(res.data(), o.p, (size_t)K * 4, hipMemcpyDeviceToHost));
    if (!rap) { *decoded = res[0]; return AOCLGPU_OK; }
    for (uint32_t k = 0; k < K; ++k) if (res[k] < 0 || (uint32_t)res[k] != dcap[k]) { *decoded = -(int64_t)(k + 1); return AOCLGPU_OK; }
    *decoded = (int64_t)total;
    return AOCLGPU_OK;
}

// ------------------------------------------------------------------------------------------------
// Host-buffer forms: what the reference's AOCL_ENABLE_THREADS build does inside its one-shot calls (partition,
// compress the partitions in parallel, join into a RAP stream / split and decode in parallel), with GPU chunks
// as the partitions.  Used by aocl_llc_compress / aocl_llc_decompress (csrc/aocl_api.cpp) when the caller asks for
// more than one thread, and by the bench CLI.
// ------------------------------------------------------------------------------------------------
extern "C" int aoclgpu_lz4_compress_chunks(const aoclgpu_batch* b, void* stream);
extern "C" int aoclgpu_snappy_compress_chunks(const aoclgpu_batch* b, void* stream);
extern "C" int aoclgpu_zstd_compress_chunks(const aoclgpu_batch* b, void* stream);

extern "C" uint64_t aoclgpu_rap_default_chunk(aoclgpu_codec codec)
{   // BASELINE configs 2-4: LZ4 256 KiB, Snappy 64 KiB, Zstd 128 KiB
    return codec == AOCLGPU_LZ4 ? (256u << 10) : (codec == AOCLGPU_SNAPPY ? (64u << 10) : (128u << 10));
}

extern "C" int64_t aoclgpu_rap_compress_host_ex(aoclgpu_codec codec, const void* hsrc, size_t n, void* hdst, size_t cap,
                                                uint64_t chunk_bytes, uint32_t flags);
extern "C" int64_t aoclgpu_rap_compress_host(aoclgpu_codec codec, const void* hsrc, size_t n, void* hdst, size_t cap,
                                             uint64_t chunk_bytes)
{
    return aoclgpu_rap_compress_host_ex(codec, hsrc, n, hdst, cap, chunk_bytes, 0);
}
namespace {
// Device buffers and streams of the host-buffer calls, kept between calls (one set per process, for whichever device the
// host entry points use; a second caller at the same time works on buffers of its own).  hipMalloc / hipFree of gigabyte
// buffers cost more than the kernels they feed, and hipFree synchronises the device.
// A slab's kernel lasts as long as its slowest chunk (~10 ms for 256 KiB of LZ4) whatever the slab's size, and the runtime
// runs four kernels side by side (its default number of hardware queues; tools/stream_overlap_probe.cpp): a call is cut into
// at most FOUR slabs, one stream each - sixteen slabs took four rounds of that time, 41 ms behind the last copy of 1 GiB.
constexpr int kHostStreams = 4;
struct HostPathCache {
    std::mutex mu;
    int dev = -1;
    void* buf[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; size_t cap[5] = {0, 0, 0, 0, 0};
    hipStream_t work[kHostStreams] = {}, copy = nullptr;
};
HostPathCache* g_hp = new HostPathCache;                       // (never destroyed: the HIP runtime may be gone before static destructors run)

// What the cache may keep between calls: the buffer set of one call on <= 1 GiB of input (the slab size of the reference's bench,
// test/codec_bench.h:58) is ~3.4 GiB; a call that needed more gives everything back when it returns, as host_release does for
// the one-shot contexts (capi.hip kHostCtxKeepBytes).  AOCLGPU_HOST_CACHE_KEEP_MB overrides (0 = keep nothing).
size_t host_cache_keep_bytes()
{
    static const size_t v = [] { const char* e = getenv("AOCLGPU_HOST_CACHE_KEEP_MB"); return e ? (size_t)atoll(e) << 20 : (size_t)4096 << 20; }();
    return v;
}
void host_cache_free_buffers(HostPathCache& c)                  // caller holds c.mu; the streams are kept (they hold no HBM)
{
    if (c.dev < 0) return;
    aoclgpu::DeviceScope ds(c.dev);
    for (int i = 0; i < 5; ++i) { if (c.buf[i]) (void)hipFree(c.buf[i]); c.buf[i] = nullptr; c.cap[i] = 0; }
}

struct HostBufs {                                             // the five buffers of one call: the cache's when it is free, else the call's own
    void* p[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    hipStream_t work[kHostStreams] = {}, copy = nullptr;
    bool cached = false;
    ~HostBufs()
    {
        // every way out of a call - the error returns too - first waits for what it put on the streams: kernels of earlier slabs
        // may still be writing these buffers, and the next caller gets them the moment the mutex is released
        for (hipStream_t q : work) if (q) (void)hipStreamSynchronize(q);
        if (copy) (void)hipStreamSynchronize(copy);
        if (cached) {
            size_t held = 0;
            for (size_t v : g_hp->cap) held += v;
            if (held > host_cache_keep_bytes()) host_cache_free_buffers(*g_hp);
            g_hp->mu.unlock();
            return;
        }
        for (void* q : p) if (q) (void)hipFree(q);
        for (hipStream_t q : work) if (q) (void)hipStreamDestroy(q);
        if (copy) (void)hipStreamDestroy(copy);
    }
    bool get(const size_t need[5])
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        if (g_hp->mu.try_lock()) {
            cached = true;
            HostPathCache& c = *g_hp;
            if (c.dev != dev) {                                // another device than last time: start over (rare: the host device is a process-wide setting)
                if (c.dev >= 0) {                              // the old device's buffers and streams are released THERE, not dropped
                    host_cache_free_buffers(c);
                    aoclgpu::DeviceScope ds(c.dev);
                    for (hipStream_t& q : c.work) { if (q) (void)hipStreamDestroy(q); q = nullptr; }
                    if (c.copy) (void)hipStreamDestroy(c.copy);
                }
                for (hipStream_t& q : c.work) q = nullptr;
                c.copy = nullptr; c.dev = dev;
            }
            for (int i = 0; i < 5; ++i) {
                if (c.cap[i] < need[i]) {
                    if (c.buf[i]) (void)hipFree(c.buf[i]);
                    c.buf[i] = nullptr; c.cap[i] = 0;
                    const size_t want = need[i] + need[i] / 8 + 4096;             // a little room to grow
                    if (hipMalloc(&c.buf[i], want) != hipSuccess) { (void)hipGetLastError(); if (hipMalloc(&c.buf[i], need[i] ? need[i] : 1) != hipSuccess) return false; c.cap[i] = need[i]; }
                    else c.cap[i] = want;
                }
                p[i] = c.buf[i];
            }
            for (int i = 0; i < kHostStreams; ++i) { if (!c.work[i] && hipStreamCreateWithFlags(&c.work[i], hipStreamNonBlocking) != hipSuccess) return false; work[i] = c.work[i]; }
            if (!c.copy && hipStreamCreateWithFlags(&c.copy, hipStreamNonBlocking) != hipSuccess) return false;
            copy = c.copy;
            return true;
        }
        for (int i = 0; i < 5; ++i) if (hipMalloc(&p[i], need[i] ? need[i] : 1) != hipSuccess) return false;
        for (int i = 0; i < kHostStreams; ++i) if (hipStreamCreateWithFlags(&work[i], hipStreamNonBlocking) != hipSuccess) return false;
        return hipStreamCreateWithFlags(&copy, hipStreamNonBlocking) == hipSuccess;
    }
};
}  // namespace

namespace aoclgpu {
// aoclgpu_host_trim's share of this file: the cached buffer set of the RAP host calls goes back to the device unless a call
// is using it right now (that call's own exit applies the keep limit).  Returns the bytes released.
size_t rap_host_cache_trim()
{
    if (!g_hp->mu.try_lock()) return 0;
    size_t held = 0;
    for (size_t v : g_hp->cap) held += v;
    host_cache_free_buffers(*g_hp);
    g_hp->mu.unlock();
    return held;
}
size_t rap_host_cache_bytes()                                   // (tests) what the cache holds now; ~0 = in use
{
    if (!g_hp->mu.try_lock()) return ~(size_t)0;
    size_t held = 0;
    for (size_t v : g_hp->cap) held += v;
    g_hp->mu.unlock();
    return held;
}
}  // namespace aoclgpu
extern "C" uint64_t aoclgpu_host_cache_bytes(void) { return (uint64_t)aoclgpu::rap_host_cache_bytes(); }

// flags: the codec's AOCLGPU_FLAG_* / AOCLGPU_ZSTD_LEVEL(l) / AOCLGPU_LZ4_ACCELERATION(a) for every partition.
// The input goes to the device in up to four slabs (whole chunks) and every slab's chunks are compressed as soon as they have
// arrived - each slab on a stream of its own, so the kernels overlap each other and the copies behind them; the
// partitions are then joined on the device (rap_pack) and the stream comes back in one copy.  (Pageable host memory moves at
// ~55 GB/s on this platform, pinned staging would not be faster - tools/host_copy_probe.cpp - so the copies are plain.)
extern "C" int64_t aoclgpu_rap_compress_host_ex(aoclgpu_codec codec, const void* hsrc, size_t n, void* hdst, size_t cap,
                                                uint64_t chunk_bytes, uint32_t flags)
{
    if ((!hsrc && n) || !hdst || n > 0x7FFFFFFFull) { set_error_msg("rap_compress_host: bad arguments"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_ARG; }
    if (chunk_bytes == 0) {
        chunk_bytes = aoclgpu_rap_default_chunk(codec);
        // The caller left the partition size to the library.  A slab's kernel lasts as long as its slowest chunk, which for LZ4 is
        // proportional to the chunk size: with 128 KiB partitions the kernels finish 8 ms behind the last copy of a 1 GiB input
        // instead of 15 ms (26.7 instead of 22.4 GiB/s end to end) and the stream grows by 0.8 %.
        if (codec == AOCLGPU_LZ4 && n >= ((size_t)64 << 20)) chunk_bytes = 128u << 10;
        if (const char* e = getenv("AOCLGPU_RAP_HOST_CHUNK_KB")) { const long kb = atol(e); if (kb >= 4) chunk_bytes = (uint64_t)kb << 10; }
    }
    if (codec == AOCLGPU_ZSTD && chunk_bytes > (128u << 10)) chunk_bytes = 128u << 10;
    uint64_t K = n ? (n + chunk_bytes - 1) / chunk_bytes : 1;
    while (K > 65535) { chunk_bytes *= 2; K = (n + chunk_bytes - 1) / chunk_bytes; }   // 16-bit partition count
    if (codec == AOCLGPU_ZSTD && chunk_bytes > (128u << 10)) { set_error_msg("rap_compress_host: zstd input needs more than 65535 partitions"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_ARG; }
    const uint64_t slot = (aoclgpu_compress_bound(codec, chunk_bytes) + 15) & ~(uint64_t)15;
    const uint64_t scap = aoclgpu_rap_frame_bytes(codec, (uint32_t)K) + K * slot + 16;
    aoclgpu::DeviceScope devScope(aoclgpu_host_device());
    if (!devScope.ok) { set_error_msg("rap_compress_host: hipSetDevice"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_NO_DEVICE; }
    const bool trace = getenv("AOCLGPU_HOST_TRACE") != nullptr;   // stage times of this call on stderr (development aid)
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    HostBufs hb;
    const size_t need[5] = {n, (size_t)(K * slot), (size_t)(K * 4), (size_t)(K * 4), (size_t)scap};
    if (!hb.get(need)) { (void)hipGetLastError(); set_error_msg("rap_compress_host: hipMalloc / hipStreamCreate"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_HIP; }
    uint8_t* dsrc = (uint8_t*)hb.p[0]; uint8_t* dslots = (uint8_t*)hb.p[1]; int32_t* dlen = (int32_t*)hb.p[2]; uint32_t* daux = (uint32_t*)hb.p[3]; uint8_t* dout = (uint8_t*)hb.p[4];
#define HCK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_error(#call, e_); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_HIP; } } while (0)
    const double t1 = now();
    uint64_t nslab = n / ((size_t)32 << 20);                                        // slabs of at least 32 MiB, at most one per stream
    if (nslab < 1) nslab = 1;
    if (nslab > (uint64_t)kHostStreams) nslab = kHostStreams;
    const uint64_t per = (K + nslab - 1) / nslab;                                   // chunks per slab
    int si = 0;
    for (uint64_t c0 = 0; c0 < K; c0 += per, ++si) {
        const uint64_t cn = K - c0 < per ? K - c0 : per;
        const uint64_t o = c0 * chunk_bytes, bytes = (o + cn * chunk_bytes <= n ? cn * chunk_bytes : n - o);
        if (bytes) {
            HCK(hipMemcpyAsync(dsrc + o, (const uint8_t*)hsrc + o, bytes, hipMemcpyHostToDevice, hb.copy));
            HCK(hipStreamSynchronize(hb.copy));
        }
        aoclgpu_batch b; memset(&b, 0, sizeof(b));
        b.src = dsrc + o; b.dst = dslots + c0 * slot; b.out_len = dlen + c0; b.aux = daux + c0;
        b.nchunks = (uint32_t)cn; b.total_bytes = n ? bytes : 0; b.chunk_bytes = chunk_bytes; b.slot_bytes = slot; b.reserved = flags;
        void* st = hb.work[si % kHostStreams];
        const int rc = codec == AOCLGPU_LZ4 ? aoclgpu_lz4_compress_chunks(&b, st)
                     : codec == AOCLGPU_SNAPPY ? aoclgpu_snappy_compress_chunks(&b, st) : aoclgpu_zstd_compress_chunks(&b, st);
        if (rc) { for (hipStream_t q : hb.work) (void)hipStreamSynchronize(q); return AOCLGPU_HOST_FAIL + rc; }
    }
    const double t2 = now();
    for (hipStream_t q : hb.work) HCK(hipStreamSynchronize(q));
    const double t3 = now();
    aoclgpu_batch b; memset(&b, 0, sizeof(b));
    b.src = dsrc; b.dst = dslots; b.out_len = dlen; b.aux = daux;
    b.nchunks = (uint32_t)K; b.total_bytes = n; b.chunk_bytes = chunk_bytes; b.slot_bytes = slot; b.reserved = flags;
    uint64_t bytes = 0;
    const int rc = aoclgpu_rap_pack(codec, &b, dout, scap, &bytes, hb.work[0]);
    if (rc) return AOCLGPU_HOST_FAIL + rc;
    if (bytes > cap) return 0;                                   // does not fit the caller's buffer
    HCK(hipStreamSynchronize(hb.work[0]));
    const double t4 = now();
    HCK(hipMemcpyAsync(hdst, dout, bytes, hipMemcpyDeviceToHost, hb.copy));
    HCK(hipStreamSynchronize(hb.copy));
    if (trace) fprintf(stderr, "rap_compress_host: buffers %.1f ms, slabs in + launches %.1f, kernels' tail %.1f, join %.1f, out %.1f (n %zu, %llu partitions)\n",
                       t1 - t0, t2 - t1, t3 - t2, t4 - t3, now() - t4, n, (unsigned long long)K);
    return (int64_t)bytes;
}

extern "C" int64_t aoclgpu_rap_decompress_host(aoclgpu_codec codec, const void* hsrc, size_t n, void* hdst, size_t cap)
{
    if (!hsrc || !hdst || n == 0) { set_error_msg("rap_decompress_host: bad arguments"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_ARG; }
    aoclgpu::DeviceScope devScope(aoclgpu_host_device());
    if (!devScope.ok) { set_error_msg("rap_decompress_host: hipSetDevice"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_NO_DEVICE; }
    HostBufs hb;
    const size_t need[5] = {n, cap, 0, 0, 0};
    if (!hb.get(need)) { (void)hipGetLastError(); set_error_msg("rap_decompress_host: hipMalloc / hipStreamCreate"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_HIP; }
    HCK(hipMemcpyAsync(hb.p[0], hsrc, n, hipMemcpyHostToDevice, hb.copy));
    HCK(hipStreamSynchronize(hb.copy));
    int64_t r = -1;
    const int rc = aoclgpu_rap_unpack(codec, (const uint8_t*)hb.p[0], n, (uint8_t*)hb.p[1], cap, &r, hb.work[0]);
    if (rc) return AOCLGPU_HOST_FAIL + rc;
    if (r > 0) { HCK(hipMemcpyAsync(hdst, hb.p[1], (size_t)r, hipMemcpyDeviceToHost, hb.copy)); HCK(hipStreamSynchronize(hb.copy)); }
#undef HCK
    return r;
}

// ------------------------------------------------------------------------------------------------
// Zstd streams of several frames on host buffers (include/aocl_gpu_codec.h).  The one-shot Zstd kernel makes one
// single-block frame per chunk (<= 128 KiB); a larger input becomes a sequence of such frames - what the reference's
// decoder, and this one, read as one stream (zstd_decompress.c:1113-1200 loops over frames).
// ------------------------------------------------------------------------------------------------
extern "C" int aoclgpu_compact_slots(const uint8_t* slots, uint64_t slot_bytes, const int32_t* out_len, uint32_t nchunks,
                                     uint8_t* packed, uint64_t* out_off, void* stream);

extern "C" int64_t aoclgpu_zstd_compress_host_frames(const void* hsrc, size_t n, void* hdst, size_t cap, uint32_t flags)
{
    const uint64_t chunk = 128u << 10;
    if ((!hsrc && n) || !hdst) { set_error_msg("zstd frames: bad arguments"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_ARG; }
    if (n <= chunk) return aoclgpu_compress_host_ex(AOCLGPU_ZSTD, hsrc, n, hdst, cap, flags);
    const uint64_t K = (n + chunk - 1) / chunk;
    if (K > 0xFFFFFFFFull) { set_error_msg("zstd frames: input too large"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_ARG; }
    const uint64_t slot = (aoclgpu_compress_bound(AOCLGPU_ZSTD, chunk) + 15) & ~(uint64_t)15;
    aoclgpu::DeviceScope devScope(aoclgpu_host_device());
    if (!devScope.ok) { set_error_msg("zstd frames: hipSetDevice"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_NO_DEVICE; }
    DevBuf dsrc, dslots, dlen, doff, dout;
    if (!dsrc.alloc(n) || !dslots.alloc(K * slot) || !dlen.alloc(K * 4) || !doff.alloc((K + 1) * 8) || !dout.alloc(K * slot)) {
        set_error_msg("zstd frames: hipMalloc"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_HIP;
    }
#define HCK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_error(#call, e_); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_HIP; } } while (0)
    HCK(hipMemcpy(dsrc.p, hsrc, n, hipMemcpyHostToDevice));
    aoclgpu_batch b; memset(&b, 0, sizeof(b));
    b.src = (const uint8_t*)dsrc.p; b.dst = (uint8_t*)dslots.p; b.out_len = (int32_t*)dlen.p;
    b.nchunks = (uint32_t)K; b.total_bytes = n; b.chunk_bytes = chunk; b.slot_bytes = slot; b.reserved = flags;
    int rc = aoclgpu_zstd_compress_chunks(&b, nullptr);
    if (rc) return AOCLGPU_HOST_FAIL + rc;
    rc = aoclgpu_compact_slots((const uint8_t*)dslots.p, slot, (const int32_t*)dlen.p, (uint32_t)K, (uint8_t*)dout.p, (uint64_t*)doff.p, nullptr);
    if (rc) return AOCLGPU_HOST_FAIL + rc;
    uint64_t total = 0;
    HCK(hipMemcpy(&total, (const uint64_t*)doff.p + K, 8, hipMemcpyDeviceToHost));
    std::vector<int32_t> lens(K);
    HCK(hipMemcpy(lens.data(), dlen.p, K * 4, hipMemcpyDeviceToHost));
    for (uint64_t k = 0; k < K; ++k) if (lens[k] <= 0) { set_error_msg("zstd frames: a chunk failed to compress"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_HIP; }
    if (total > cap) return 0;
    HCK(hipMemcpy(hdst, dout.p, total, hipMemcpyDeviceToHost));
    return (int64_t)total;
}

namespace {
// Frame boundaries from the frame and block headers alone (ZSTD_findFrameCompressedSize, zstd_decompress.c:700-790).
// content = -1 when the frame header carries no content size; ck = the stored checksum when the frame has one.
struct FrameSpan { uint64_t off; uint32_t len; int64_t content; bool hasCk; uint32_t ck; };
bool split_frames(const uint8_t* p, size_t n, std::vector<FrameSpan>& out)
{
    size_t pos = 0;
    while (pos < n) {
        if (n - pos < 8) return false;
        uint32_t magic; memcpy(&magic, p + pos, 4);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
            uint32_t sz; memcpy(&sz, p + pos + 4, 4);
            if ((uint64_t)sz + 8 > n - pos) return false;
            pos += 8 + (size_t)sz; continue;
        }
        if (magic != 0xFD2FB528u) return false;
        const size_t start = pos;
        const uint32_t fhd = p[pos + 4], fcsId = fhd >> 6, single = (fhd >> 5) & 1, hasCk = (fhd >> 2) & 1, did = fhd & 3;
        pos += 5;
        if (!single) pos += 1;
        pos += did == 3 ? 4 : did;
        const size_t fl = fcsId == 0 ? (single ? 1 : 0) : (fcsId == 1 ? 2 : (fcsId == 2 ? 4 : 8));
        if (pos + fl > n) return false;
        int64_t content = -1;
        if (fl) {
            uint64_t fcs = 0;
            for (size_t i = 0; i < fl; ++i) fcs |= (uint64_t)p[pos + i] << (8 * i);
            if (fl == 2) fcs += 256;
            if (fcs > 0x7FFFFFFFull) return false;
            content = (int64_t)fcs;
        }
        pos += fl;
        for (;;) {
            if (n - pos < 3) return false;
            const uint32_t bh = p[pos] | ((uint32_t)p[pos + 1] << 8) | ((uint32_t)p[pos + 2] << 16);
            const uint32_t last = bh & 1, type = (bh >> 1) & 3, bsz = bh >> 3;
            pos += 3;
            if (type == 3) return false;
            const size_t body = type == 1 ? 1 : bsz;
            if (body > n - pos) return false;
            pos += body;
            if (last) break;
        }
        uint32_t ck = 0;
        if (hasCk) { if (n - pos < 4) return false; memcpy(&ck, p + pos, 4); pos += 4; }
        if (pos - start > 0xFFFFFFFFull) return false;
        out.push_back(FrameSpan{start, (uint32_t)(pos - start), content, hasCk != 0, ck});
    }
    return true;
}

// XXH64 (the published xxHash algorithm, seed 0): Zstandard's content checksum is its low 32 bits (zstd_decompress.c:1087-1094)
uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
uint64_t xxh64(const uint8_t* p, size_t len)
{
    const uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full, P3 = 0x165667B19E3779F9ull, P4 = 0x85EBCA77C2B2AE63ull, P5 = 0x27D4EB2F165667C5ull;
    auto rd64 = [](const uint8_t* q) { uint64_t v; memcpy(&v, q, 8); return v; };
    auto rd32 = [](const uint8_t* q) { uint32_t v; memcpy(&v, q, 4); return v; };
    auto round = [&](uint64_t acc, uint64_t in) { return rotl64(acc + in * P2, 31) * P1; };
    auto merge = [&](uint64_t h, uint64_t v) { return (h ^ round(0, v)) * P1 + P4; };
    const uint8_t* const end = p + len;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = P1 + P2, v2 = P2, v3 = 0, v4 = 0 - P1;
        do { v1 = round(v1, rd64(p)); v2 = round(v2, rd64(p + 8)); v3 = round(v3, rd64(p + 16)); v4 = round(v4, rd64(p + 24)); p += 32; } while (p + 32 <= end);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = merge(h, v1); h = merge(h, v2); h = merge(h, v3); h = merge(h, v4);
    } else h = P5;
    h += (uint64_t)len;
    while (p + 8 <= end) { h ^= round(0, rd64(p)); h = rotl64(h, 27) * P1 + P4; p += 8; }
    if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * P1; h = rotl64(h, 23) * P2 + P3; p += 4; }
    while (p < end) { h ^= (uint64_t)(*p) * P5; h = rotl64(h, 11) * P1; p++; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

// content checksums of a decoded stream: true when every frame that carries one matches (frames are laid out back to
// back in dst; a frame without a content size can only be checked when it is the stream's only frame)
bool checksums_ok(const std::vector<FrameSpan>& fr, const uint8_t* dst, size_t total)
{
    size_t o = 0;
    for (size_t k = 0; k < fr.size(); ++k) {
        size_t len;
        if (fr[k].content >= 0) len = (size_t)fr[k].content;
        else if (fr.size() == 1) len = total;
        else return true;                                      // cannot locate the frames behind this one: not checked
        if (o + len > total) return false;
        if (fr[k].hasCk && (uint32_t)xxh64(dst + o, len) != fr[k].ck) return false;
        o += len;
    }
    return true;
}
}  // namespace

extern "C" int64_t aoclgpu_zstd_decompress_host_frames(const void* hsrc, size_t n, void* hdst, size_t cap)
{
    if (!hsrc || !hdst) { set_error_msg("zstd frames: bad arguments"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_ARG; }
    std::vector<FrameSpan> fr;
    const bool split = n <= 0xFFFFFFFFull && split_frames((const uint8_t*)hsrc, n, fr);
    bool sized = split && fr.size() >= 2;
    if (sized) for (const FrameSpan& f : fr) sized = sized && f.content >= 0;
    if (!sized) {                                             // one wavefront walks the stream
        // (no shortcut for a content size that exceeds `cap`: the reference decodes until the output runs out, and whatever is
        //  wrong with the stream in front of that point is what it reports)
        const int64_t r = aoclgpu_decompress_host(AOCLGPU_ZSTD, hsrc, n, hdst, cap);
        if (r >= 0 && split && !checksums_ok(fr, (const uint8_t*)hdst, (size_t)r)) return -22;   // checksum_wrong (zstd_decompress.c:1101)
        return r;
    }
    const size_t K = fr.size();
    uint64_t total = 0;
    std::vector<uint64_t> soff(K), doff(K); std::vector<uint32_t> slen(K), dcap(K);
    // a frame gets the room the frames in front of it leave (the reference decodes them in order: the first frame that does not
    // fit fails with dstSize_tooSmall - or with whatever is wrong with it in front of that point)
    uint64_t full = 0;
    for (size_t k = 0; k < K; ++k) {
        soff[k] = fr[k].off; slen[k] = fr[k].len; doff[k] = full < cap ? full : cap;
        const uint64_t left = cap - doff[k];
        dcap[k] = (uint32_t)((uint64_t)fr[k].content < left ? (uint64_t)fr[k].content : left);
        full += (uint64_t)fr[k].content; total += dcap[k];
    }
    aoclgpu::DeviceScope devScope(aoclgpu_host_device());
    if (!devScope.ok) { set_error_msg("zstd frames: hipSetDevice"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_NO_DEVICE; }
    DevBuf dsrc, ddst, a, bq, c, d, o;
    if (!dsrc.alloc(n) || !ddst.alloc(total) || !a.alloc(K * 8) || !bq.alloc(K * 4) || !c.alloc(K * 8) || !d.alloc(K * 4) || !o.alloc(K * 4)) {
        set_error_msg("zstd frames: hipMalloc"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_HIP;
    }
    HCK(hipMemcpy(dsrc.p, hsrc, n, hipMemcpyHostToDevice));
    HCK(hipMemcpy(a.p, soff.data(), K * 8, hipMemcpyHostToDevice));
    HCK(hipMemcpy(bq.p, slen.data(), K * 4, hipMemcpyHostToDevice));
    HCK(hipMemcpy(c.p, doff.data(), K * 8, hipMemcpyHostToDevice));
    HCK(hipMemcpy(d.p, dcap.data(), K * 4, hipMemcpyHostToDevice));
    aoclgpu_batch bt; memset(&bt, 0, sizeof(bt));
    bt.src = (const uint8_t*)dsrc.p; bt.dst = (uint8_t*)ddst.p; bt.src_off = (const uint64_t*)a.p; bt.src_len = (const uint32_t*)bq.p;
    bt.dst_off = (const uint64_t*)c.p; bt.dst_cap = (const uint32_t*)d.p; bt.out_len = (int32_t*)o.p; bt.nchunks = (uint32_t)K;
    const int rc = aoclgpu_zstd_decompress_chunks(&bt, nullptr);
    if (rc) return AOCLGPU_HOST_FAIL + rc;
    std::vector<int32_t> res(K);
    HCK(hipMemcpy(res.data(), o.p, K * 4, hipMemcpyDeviceToHost));
    if (total) HCK(hipMemcpy(hdst, ddst.p, total, hipMemcpyDeviceToHost));
#undef HCK
    for (size_t k = 0; k < K; ++k) {
        if (res[k] < 0 || (uint64_t)res[k] != (uint64_t)fr[k].content) {
            // A frame failed.  WHICH error the reference reports depends on what it had decoded when it noticed: it gives every
            // frame all the room that is left (a frame whose header understates its size runs on to its end and is then
            // corruption_detected, not dstSize_tooSmall) - so the stream is walked once more by one wavefront, in the reference's
            // order with the reference's capacities; its verdict is the return value.  (Cold path: errors only.)
            const int64_t r = aoclgpu_decompress_host(AOCLGPU_ZSTD, hsrc, n, hdst, cap);
            if (r >= 0 && !checksums_ok(fr, (const uint8_t*)hdst, (size_t)r)) return -22;
            return r >= 0 ? (res[k] < 0 ? res[k] : -20) : r;    // (a walk that finds nothing wrong leaves the parallel decode's verdict)
        }
        if (fr[k].hasCk && (uint32_t)xxh64((const uint8_t*)hdst + doff[k], (size_t)fr[k].content) != fr[k].ck) return -22;   // checksum_wrong (zstd_decompress.c:1101)
    }
    return (int64_t)total;
}

#define HCK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_error(#call, e_); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_HIP; } } while (0)
// ONE frame for an input of any size below 2 GiB with explicit "fast" or "double-fast" parameters: host buffers in and out around
// aoclgpu_zstd_compress_frames.  checksumFlag: the frame ends with the low 32 bits of XXH64(src) and says so in its header
// (ZSTD_writeEpilogue, zstd_compress.c:4700-4706).  Returns the frame size, 0 when it does not fit `cap`, negative on failure.
extern "C" int64_t aoclgpu_zstd_compress_host_frame(const void* hsrc, size_t n, void* hdst, size_t cap,
                                                    uint32_t strategy, uint32_t windowLog, uint32_t hashLog, uint32_t chainLog,
                                                    uint32_t minMatch, uint32_t targetLength, uint32_t checksumFlag)
{
    if ((!hsrc && n) || !hdst || n >= ((size_t)1 << 31)) { set_error_msg("zstd frame: bad arguments"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_ARG; }
    aoclgpu::DeviceScope devScope(aoclgpu_host_device());
    if (!devScope.ok) { set_error_msg("zstd frame: hipSetDevice"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_NO_DEVICE; }
    const size_t tail = checksumFlag ? 4 : 0;
    if (cap < tail) return 0;
    const size_t bound = aoclgpu_compress_bound(AOCLGPU_ZSTD, n) + 32, room = cap - tail < bound ? cap - tail : bound;
    DevBuf dsrc, ddst, dmeta;
    if (!dsrc.alloc(n ? n : 1) || !ddst.alloc(room ? room : 1) || !dmeta.alloc(5 * 8)) { set_error_msg("zstd frame: hipMalloc"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_HIP; }
    const uint64_t meta[5] = { 0, n, 0, room, 0 };                       // src_off[2], dst_off[2], out_len
    if (n) HCK(hipMemcpy(dsrc.p, hsrc, n, hipMemcpyHostToDevice));
    HCK(hipMemcpy(dmeta.p, meta, sizeof(meta), hipMemcpyHostToDevice));
    aoclgpu_zstd_frames f; memset(&f, 0, sizeof(f));
    f.src = (const uint8_t*)dsrc.p; f.dst = (uint8_t*)ddst.p;
    f.src_off = (const uint64_t*)dmeta.p; f.dst_off = (const uint64_t*)dmeta.p + 2; f.out_len = (int64_t*)dmeta.p + 4;
    f.nframes = 1; f.windowLog = windowLog; f.hashLog = hashLog; f.minMatch = minMatch; f.targetLength = targetLength;
    f.strategy = strategy; f.chainLog = chainLog;
    const int rc = aoclgpu_zstd_compress_frames(&f, nullptr);
    if (rc) return AOCLGPU_HOST_FAIL + rc;
    int64_t r = 0;
    HCK(hipMemcpy(&r, (const int64_t*)dmeta.p + 4, 8, hipMemcpyDeviceToHost));
    if (r <= 0) return 0;
    HCK(hipMemcpy(hdst, ddst.p, (size_t)r, hipMemcpyDeviceToHost));
    if (checksumFlag) {
        const uint32_t ck = (uint32_t)xxh64((const uint8_t*)hsrc, n);
        ((uint8_t*)hdst)[4] |= 4;                                        // Content_Checksum_flag of the frame header descriptor
        memcpy((uint8_t*)hdst + r, &ck, 4);
        r += 4;
    }
    return r;
}


// Zstd RAP stream in the REFERENCE'S partition geometry, host buffers: what ZSTD_compress_advanced of the AOCL_ENABLE_THREADS build
// writes when `nparts` threads take part (zstd_compress.c:5410-5522): partition t covers src[t * (n / nparts), ...) and the last
// one takes the n % nparts leftover bytes (threads/threads.c:121-136), every partition becomes one frame with the caller's
// compression parameters (one wavefront each, side by side), the frames travel behind the RAP frame in a skippable frame.
// nparts = 1: a plain frame, no RAP frame (:5429-5437).  Returns the stream size, 0 when it does not fit, < 0 on failure.
extern "C" int64_t aoclgpu_zstd_rap_compress_host_parts(const void* hsrc, size_t n, void* hdst, size_t cap, uint32_t nparts,
                                                        uint32_t strategy, uint32_t windowLog, uint32_t hashLog, uint32_t chainLog,
                                                        uint32_t minMatch, uint32_t targetLength)
{
    if ((!hsrc && n) || !hdst || n >= ((size_t)1 << 31) || nparts == 0 || nparts > 65535 || nparts > (n ? n : 1)) {
        set_error_msg("zstd rap parts: bad arguments"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_ARG;
    }
    if (nparts == 1) return aoclgpu_zstd_compress_host_frame(hsrc, n, hdst, cap, strategy, windowLog, hashLog, chainLog, minMatch, targetLength, 0);
    aoclgpu::DeviceScope devScope(aoclgpu_host_device());
    if (!devScope.ok) { set_error_msg("zstd rap parts: hipSetDevice"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_NO_DEVICE; }
    const uint32_t K = nparts;
    const uint64_t common = n / K, left = n % K;
    std::vector<uint64_t> soff(K + 1), doff(K + 1);
    std::vector<uint32_t> slen(K);
    uint64_t dpos = 0;
    for (uint32_t k = 0; k < K; ++k) {
        soff[k] = common * k; slen[k] = (uint32_t)(common + (k == K - 1 ? left : 0));
        doff[k] = dpos; dpos += (aoclgpu_compress_bound(AOCLGPU_ZSTD, slen[k]) + 32 + 15) & ~(uint64_t)15;
    }
    soff[K] = n; doff[K] = dpos;
    const uint64_t scap = aoclgpu_rap_frame_bytes(AOCLGPU_ZSTD, K) + dpos + 64;
    DevBuf dsrc, dfr, dmeta, dout;
    if (!dsrc.alloc(n) || !dfr.alloc(dpos) || !dmeta.alloc((size_t)(3 * K + 2) * 8) || !dout.alloc(scap)) { set_error_msg("zstd rap parts: hipMalloc"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_HIP; }
    HCK(hipMemcpy(dsrc.p, hsrc, n, hipMemcpyHostToDevice));
    HCK(hipMemcpy(dmeta.p, soff.data(), (size_t)(K + 1) * 8, hipMemcpyHostToDevice));
    HCK(hipMemcpy((uint64_t*)dmeta.p + K + 1, doff.data(), (size_t)(K + 1) * 8, hipMemcpyHostToDevice));
    aoclgpu_zstd_frames f; memset(&f, 0, sizeof(f));
    f.src = (const uint8_t*)dsrc.p; f.dst = (uint8_t*)dfr.p;
    f.src_off = (const uint64_t*)dmeta.p; f.dst_off = (const uint64_t*)dmeta.p + K + 1; f.out_len = (int64_t*)dmeta.p + 2 * K + 2;
    f.nframes = K; f.windowLog = windowLog; f.hashLog = hashLog; f.minMatch = minMatch; f.targetLength = targetLength;
    f.strategy = strategy; f.chainLog = chainLog;
    int rc = aoclgpu_zstd_compress_frames(&f, nullptr);
    if (rc) return AOCLGPU_HOST_FAIL + rc;
    std::vector<int64_t> ol(K);
    HCK(hipMemcpy(ol.data(), (const int64_t*)dmeta.p + 2 * K + 2, (size_t)K * 8, hipMemcpyDeviceToHost));
    std::vector<int32_t> clen(K);
    for (uint32_t k = 0; k < K; ++k) { if (ol[k] <= 0) { set_error_msg("zstd rap parts: a partition did not compress"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_ARG; } clen[k] = (int32_t)ol[k]; }
    uint64_t bytes = 0;
    rc = aoclgpu_rap_assemble(AOCLGPU_ZSTD, (const uint8_t*)dfr.p, doff.data(), clen.data(), slen.data(), nullptr, K, (uint8_t*)dout.p, scap, &bytes, nullptr);
    if (rc) return AOCLGPU_HOST_FAIL + rc;
    if (bytes > cap) return 0;
    HCK(hipMemcpy(hdst, dout.p, bytes, hipMemcpyDeviceToHost));
    return (int64_t)bytes;
}

#undef HCK
