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
#include <vector>
#include "common.hip.h"
#include "host_util.h"

namespace aoclgpu {
namespace {

constexpr uint64_t kRapMagic = 0x434C4C5F4C434F41ull;       // "AOCL_LLC" (threads/threads.h:79)
constexpr uint32_t kRapHead = 8 + 4 + 2 + 2;                // magic, frame length, main / child thread counts
constexpr uint32_t kRapEntry = 12;                          // offset, length, decoded length
constexpr uint32_t kZstdSkipMagic = 0x184D2A50u;            // ZSTD_MAGIC_SKIPPABLE_START, variant 0 (zstd_compress.c:5353)
constexpr uint32_t kHead = 16;                              // bytes of every slot fetched for the first-token parse

struct Piece {                                              // one partition of the output stream
    uint64_t dst;                                           // offset in the stream
    uint64_t lit_src;                                       // offset in the uncompressed source
    uint64_t body_src;                                      // offset in the slot buffer
    uint32_t hdr_off, hdr_len;                              // rewritten first-sequence header, in the header blob
    uint32_t lit_len, body_len;
};

__global__ void __launch_bounds__(64) rap_gather_kernel(const Piece* __restrict__ pieces, uint32_t n,
                                                        const uint8_t* __restrict__ blob, const uint8_t* __restrict__ src,
                                                        const uint8_t* __restrict__ slots, uint8_t* __restrict__ out)
{
    const uint32_t k = blockIdx.x;
    if (k >= n) return;
    const int lane = lane_id();
    const Piece p = pieces[k];
    gbyte* d = (gbyte*)out + p.dst;
    wave_copy(d, (const gbyte*)blob + p.hdr_off, p.hdr_len, lane);
    wave_copy(d + p.hdr_len, (const gbyte*)src + p.lit_src, p.lit_len, lane);
    wave_copy(d + p.hdr_len + p.lit_len, (const gbyte*)slots + p.body_src, p.body_len, lane);
}

__global__ void rap_heads_kernel(const uint8_t* __restrict__ slots, const uint64_t* __restrict__ slot_off, uint64_t slot_bytes,
                                 uint32_t n, uint8_t* __restrict__ heads)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint8_t* s = slots + (slot_off ? slot_off[k] : (uint64_t)k * slot_bytes);
    for (uint32_t i = 0; i < kHead; ++i) heads[(uint64_t)k * kHead + i] = s[i];   // slots are >= 16 bytes (compress bounds)
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
    uint64_t total_src = 0;
    for (uint32_t k = 0; k < K; ++k) {
        if (olen[k] <= 0) { set_error_msg("rap_pack: a chunk failed to compress"); return AOCLGPU_ERR_ARG; }
        total_src += n[k];
    }
    if (total_src > 0x7FFFFFFFull) { set_error_msg("rap_pack: RAP offsets are 32-bit, pack at most 2 GiB per stream"); return AOCLGPU_ERR_ARG; }

    std::vector<Piece> pieces(K);
    std::vector<uint8_t> blob;                                   // frame + rewritten headers
    const uint32_t frame = (uint32_t)aoclgpu_rap_frame_bytes(codec, K);
    blob.resize(frame, 0);
    uint64_t pos = frame;                                        // running output offset
    const uint32_t base = (codec == AOCLGPU_ZSTD && K > 1) ? 8u : 0u;   // RAP offsets count from the magic word
    auto entry = [&](uint32_t k, uint64_t off, uint32_t len, uint32_t dlen) {
        uint8_t* e = blob.data() + base + kRapHead + (size_t)kRapEntry * k;
        put32(e, (uint32_t)(off - base)); put32(e + 4, len); put32(e + 8, dlen);
    };
    if (K == 1) {                                                // one partition: the plain one-shot stream
        pieces[0] = Piece{0, 0, slotoff[0], 0, 0, 0, (uint32_t)olen[0]};
        pos = (uint64_t)olen[0];
    } else if (codec == AOCLGPU_ZSTD) {
        for (uint32_t k = 0; k < K; ++k) {
            pieces[k] = Piece{pos, 0, slotoff[k], 0, 0, 0, (uint32_t)olen[k]};
            entry(k, pos, (uint32_t)olen[k], n[k]);
            pos += (uint32_t)olen[k];
        }
    } else if (codec == AOCLGPU_SNAPPY) {
        const uint32_t hoff = (uint32_t)blob.size();
        put_varint(blob, (uint32_t)total_src);                   // one length prefix for the whole stream
        const uint32_t hlen = (uint32_t)blob.size() - hoff;
        for (uint32_t k = 0; k < K; ++k) {
            const uint32_t w = varint_len(n[k]);
            if ((uint32_t)olen[k] < w) { set_error_msg("rap_pack: snappy chunk shorter than its prefix"); return AOCLGPU_ERR_ARG; }
            const uint32_t body = (uint32_t)olen[k] - w, hl = k == 0 ? hlen : 0;
            pieces[k] = Piece{pos, 0, slotoff[k] + w, k == 0 ? hoff : 0, hl, 0, body};
            entry(k, pos + hl, body, n[k]);
            pos += hl + body;
        }
    } else {                                                     // LZ4 (lz4.c:2736-2906)
        std::vector<uint8_t> heads((size_t)K * kHead);
        {
            DevBuf dh, doff;
            if (!dh.alloc((size_t)K * kHead)) { set_error_msg("rap_pack: hipMalloc"); return AOCLGPU_ERR_HIP; }
            const uint64_t* dso = nullptr;
            if (b->src_off) dso = b->dst_off;
            hipLaunchKernelGGL(rap_heads_kernel, dim3((K + 255) / 256), dim3(256), 0, s, b->dst, dso, b->slot_bytes, K, (uint8_t*)dh.p);
            if (int rc = check_launch("rap_heads_kernel")) return rc;
            HIPCK(hipStreamSynchronize(s));
            HIPCK(hipMemcpy(heads.data(), dh.p, heads.size(), hipMemcpyDeviceToHost));
        }
        uint64_t pend = 0;                                       // literals waiting for the next partition's first sequence
        uint32_t prev_len = 0; uint64_t prev_off = frame;
        for (uint32_t k = 0; k < K; ++k) {
            const bool last = k + 1 == K;
            const uint32_t L = last ? 0 : aux[k];
            const uint32_t lastseq = last ? 0 : 1 + (L < 15 ? 0 : 1 + (L - 15) / 255) + L;
            if ((uint32_t)olen[k] < lastseq || L > n[k]) { set_error_msg("rap_pack: inconsistent aux[] (last literal run)"); return AOCLGPU_ERR_ARG; }
            const uint32_t body_end = (uint32_t)olen[k] - lastseq;      // dst_trap_size of the reference's MT compressor
            const uint64_t off = prev_off + prev_len;
            if (k == 0) {
                pieces[0] = Piece{pos, 0, slotoff[0], 0, 0, 0, body_end};
                entry(0, pos, body_end, n[0] - L);
                prev_off = pos; prev_len = body_end; pos += body_end; pend = L;
                continue;
            }
            if (!last && body_end == 0) {                        // all literals: they travel on to the next partition (:2808-2823)
                pieces[k] = Piece{pos, 0, 0, 0, 0, 0, 0};
                entry(k, off, 0, 0);
                prev_off = off; prev_len = 0; pend += n[k];
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
                        if (!big.empty() || (uint32_t)olen[k] <= kHead) { set_error_msg("rap_pack: truncated first token"); return AOCLGPU_ERR_ARG; }
                        big.resize((uint32_t)olen[k] < 4096u ? (uint32_t)olen[k] : 4096u);
                        HIPCK(hipMemcpy(big.data(), b->dst + slotoff[k], big.size(), hipMemcpyDeviceToHost));
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
            if (body_end < hdr_in) { set_error_msg("rap_pack: first token longer than the partition"); return AOCLGPU_ERR_ARG; }
            const uint32_t body = body_end - hdr_in;
            pieces[k] = Piece{pos, srcoff[k] - pend, slotoff[k] + hdr_in, hoff, hlen, (uint32_t)pend, body};
            const uint32_t plen = hlen + (uint32_t)pend + body;
            entry(k, off, plen, (uint32_t)((n[k] - L) + pend));
            prev_off = off; prev_len = plen; pos += plen; pend = L;
        }
    }
    if (K > 1) {                                                 // frame header (threads/threads.c:104-111)
        uint8_t* f = blob.data();
        if (codec == AOCLGPU_ZSTD) { put32(f, kZstdSkipMagic); put32(f + 4, frame - 8); f += 8; }
        memcpy(f, &kRapMagic, 8);
        put32(f + 8, frame - base);
        put32(f + 12, K);                                        // u16 partitions, u16 child threads = 0
    }
    if (pos > cap) { set_error_msg("rap_pack: stream_cap too small"); return AOCLGPU_ERR_ARG; }
    if (pos > 0xFFFFFFFFull) { set_error_msg("rap_pack: stream exceeds 32-bit RAP offsets"); return AOCLGPU_ERR_ARG; }
    // ---- execute: frame + three copies per partition ----
    DevBuf dp, db;
    if (!dp.alloc(pieces.size() * sizeof(Piece)) || !db.alloc(blob.size())) { set_error_msg("rap_pack: hipMalloc"); return AOCLGPU_ERR_HIP; }
    HIPCK(hipMemcpyAsync(dp.p, pieces.data(), pieces.size() * sizeof(Piece), hipMemcpyHostToDevice, s));
    if (!blob.empty()) HIPCK(hipMemcpyAsync(db.p, blob.data(), blob.size(), hipMemcpyHostToDevice, s));
    if (frame) HIPCK(hipMemcpyAsync(out, db.p, frame, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(rap_gather_kernel, dim3(K), dim3(64), 0, s, (const Piece*)dp.p, K, (const uint8_t*)db.p, b->src, b->dst, out);
    if (int rc = check_launch("rap_gather_kernel")) return rc;
    HIPCK(hipStreamSynchronize(s));                              // the plan buffers die with this call
    *stream_bytes = pos;
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

extern "C" int64_t aoclgpu_rap_compress_host(aoclgpu_codec codec, const void* hsrc, size_t n, void* hdst, size_t cap,
                                             uint64_t chunk_bytes)
{
    if ((!hsrc && n) || !hdst || n > 0x7FFFFFFFull) { set_error_msg("rap_compress_host: bad arguments"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_ARG; }
    if (chunk_bytes == 0) chunk_bytes = aoclgpu_rap_default_chunk(codec);
    if (codec == AOCLGPU_ZSTD && chunk_bytes > (128u << 10)) chunk_bytes = 128u << 10;
    uint64_t K = n ? (n + chunk_bytes - 1) / chunk_bytes : 1;
    while (K > 65535) { chunk_bytes *= 2; K = (n + chunk_bytes - 1) / chunk_bytes; }   // 16-bit partition count
    if (codec == AOCLGPU_ZSTD && chunk_bytes > (128u << 10)) { set_error_msg("rap_compress_host: zstd input needs more than 65535 partitions"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_ARG; }
    const uint64_t slot = (aoclgpu_compress_bound(codec, chunk_bytes) + 15) & ~(uint64_t)15;
    const uint64_t scap = aoclgpu_rap_frame_bytes(codec, (uint32_t)K) + K * slot + 16;
    DevBuf dsrc, dslots, dlen, daux, dout;
    if (!dsrc.alloc(n) || !dslots.alloc(K * slot) || !dlen.alloc(K * 4) || !daux.alloc(K * 4) || !dout.alloc(scap)) {
        set_error_msg("rap_compress_host: hipMalloc"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_HIP;
    }
#define HCK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_error(#call, e_); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_HIP; } } while (0)
    if (n) HCK(hipMemcpy(dsrc.p, hsrc, n, hipMemcpyHostToDevice));
    aoclgpu_batch b; memset(&b, 0, sizeof(b));
    b.src = (const uint8_t*)dsrc.p; b.dst = (uint8_t*)dslots.p; b.out_len = (int32_t*)dlen.p; b.aux = (uint32_t*)daux.p;
    b.nchunks = (uint32_t)K; b.total_bytes = n; b.chunk_bytes = chunk_bytes; b.slot_bytes = slot;
    int rc = codec == AOCLGPU_LZ4 ? aoclgpu_lz4_compress_chunks(&b, nullptr)
           : codec == AOCLGPU_SNAPPY ? aoclgpu_snappy_compress_chunks(&b, nullptr) : aoclgpu_zstd_compress_chunks(&b, nullptr);
    if (rc) return AOCLGPU_HOST_FAIL + rc;
    uint64_t bytes = 0;
    rc = aoclgpu_rap_pack(codec, &b, (uint8_t*)dout.p, scap, &bytes, nullptr);
    if (rc) return AOCLGPU_HOST_FAIL + rc;
    if (bytes > cap) return 0;                                   // does not fit the caller's buffer
    HCK(hipMemcpy(hdst, dout.p, bytes, hipMemcpyDeviceToHost));
    return (int64_t)bytes;
}

extern "C" int64_t aoclgpu_rap_decompress_host(aoclgpu_codec codec, const void* hsrc, size_t n, void* hdst, size_t cap)
{
    if (!hsrc || !hdst || n == 0) { set_error_msg("rap_decompress_host: bad arguments"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_ARG; }
    DevBuf dsrc, ddst;
    if (!dsrc.alloc(n) || !ddst.alloc(cap)) { set_error_msg("rap_decompress_host: hipMalloc"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_HIP; }
    HCK(hipMemcpy(dsrc.p, hsrc, n, hipMemcpyHostToDevice));
    int64_t r = -1;
    const int rc = aoclgpu_rap_unpack(codec, (const uint8_t*)dsrc.p, n, (uint8_t*)ddst.p, cap, &r, nullptr);
    if (rc) return AOCLGPU_HOST_FAIL + rc;
    if (r > 0) HCK(hipMemcpy(hdst, ddst.p, (size_t)r, hipMemcpyDeviceToHost));
#undef HCK
    return r;
}
