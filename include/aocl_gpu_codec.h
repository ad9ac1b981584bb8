/* include/aocl_gpu_codec.h -- C-ABI of the MI355X block-codec engine (libaocl_compression_gpu.so).
 *
 * Device-side batch entry points: every pointer below is a DEVICE pointer (HBM) unless the name
 * says `host`; sizes are plain integers; `stream` is a hipStream_t passed as void* (NULL = the
 * default stream).  No torch / C++ types cross this boundary.
 *
 * One CHUNK is one independent block in the reference's sense: it is compressed exactly as the
 * reference compresses one buffer with the codec's one-shot native API, and chunk i of a batch is
 * handled by one 64-lane wavefront with its hash table in LDS.  This is the GPU counterpart of one
 * RAP partition of the reference's OpenMP path (threads/threads.c:121-153, lz4.c:2684-2731,
 * snappy.cc:2533-2562, zstd_compress.c:5443-5489).
 *
 * What each entry point replaces in the reference:
 *   aoclgpu_lz4_compress_chunks      LZ4_compress_default            algos/lz4/lz4.c:2967-2975
 *                                    (-> AOCL_LZ4_compress_generic_validated :1371-1845)
 *   aoclgpu_lz4_decompress_chunks    LZ4_decompress_safe             algos/lz4/lz4.c:4898-4923
 *                                    (-> AOCL_LZ4_decompress_generic :3367-3803)
 *   aoclgpu_snappy_compress_chunks   snappy::RawCompress             algos/snappy/snappy.cc:2494-2666
 *                                    (-> AOCL_CompressFragment :846-1046)
 *   aoclgpu_snappy_decompress_chunks snappy::RawUncompress           algos/snappy/snappy.cc:2271-2390
 *   aoclgpu_zstd_compress_chunks     ZSTD_compress(level 1)          algos/zstd/lib/compress/zstd_compress.c:5388
 *                                    (-> zstd_fast.c:427-706 + entropy stage zstd_compress.c:2910-3064)
 *   aoclgpu_zstd_decompress_chunks   ZSTD_decompress                 algos/zstd/lib/decompress/zstd_decompress.c:1252
 *   aoclgpu_zstd_compress_frames     ZSTD_compress_advanced, any size algos/zstd/lib/compress/zstd_compress.c:5388-5470
 *                                    (-> ZSTD_compress_frameChunk :4504-4583: one frame of several blocks; fast / double-fast)
 *   aoclgpu_lz4_compress_chunks      the blocks of LZ4F_compressFrame algos/lz4/lz4frame.c:373-419, lz4.c:3179-3256
 *     with AOCLGPU_FLAG_LZ4_LINKED   with LZ4F_blockLinked (LZ4_compress_fast_continue in prefix mode)
 * Output bytes are identical to those functions' output on the same chunk / input.
 */
#ifndef AOCL_GPU_CODEC_H
#define AOCL_GPU_CODEC_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Return codes of the batch launchers (per-chunk results are reported in out_len[]). */
#define AOCLGPU_OK            0
#define AOCLGPU_ERR_ARG      -1   /* NULL / inconsistent arguments */
#define AOCLGPU_ERR_HIP      -2   /* a HIP runtime call failed (see aoclgpu_last_error) */
#define AOCLGPU_ERR_NO_DEVICE -3  /* no gfx950 device visible */

typedef enum { AOCLGPU_LZ4 = 0, AOCLGPU_SNAPPY = 1, AOCLGPU_ZSTD = 2 } aoclgpu_codec;

/* A batch of chunks.  Two addressing modes:
 *  - explicit : src_off/src_len/dst_off/dst_cap are device arrays of nchunks entries;
 *  - uniform  : src_off == NULL; chunk i covers src[i*chunk_bytes, min(total_bytes,(i+1)*chunk_bytes))
 *               and owns the output slot dst[i*slot_bytes, (i+1)*slot_bytes).
 * out_len[i] (device, int32): compress  -> compressed size, 0 if the chunk does not fit dst_cap
 *                                          (same contract as LZ4_compress_default / api/codec.cpp:126-134);
 *                             decompress-> decoded size, or a negative value for a malformed chunk. */
typedef struct {
    const uint8_t*  src;
    uint8_t*        dst;
    const uint64_t* src_off;   /* NULL => uniform mode */
    const uint32_t* src_len;
    const uint64_t* dst_off;
    const uint32_t* dst_cap;
    int32_t*        out_len;
    uint32_t        nchunks;
    uint32_t        reserved;     /* AOCLGPU_FLAG_* below.  Bits under 0x100 are ignored by the release library (a build
                                     with -DAOCLGPU_EXPERIMENTS reads them as switches of the profiling tools in tools/) */
    uint64_t        total_bytes;  /* uniform mode */
    uint64_t        chunk_bytes;  /* uniform mode */
    uint64_t        slot_bytes;   /* uniform mode */
    uint32_t*       aux;          /* optional per-chunk side output (device, nchunks entries), may be NULL:
                                     LZ4 compress -> length of the chunk's final literal run (what the reference's MT
                                     compressor returns as last_bytes_len, lz4.c:2333-2337), needed by aoclgpu_rap_pack */
} aoclgpu_batch;

/* decompress: the chunks are partitions of a RAP stream (threads/threads.h:20-75).  LZ4: every chunk but the last
 * may end right after a match (lz4.c:3935,4147-4181); Snappy: chunks carry no varint length prefix and dst_cap[i] is the
 * exact decoded size (snappy.cc:2567-2654); Zstd: no effect (partitions are whole frames). */
#define AOCLGPU_FLAG_RAP_PARTITIONS 0x100u
/* LZ4, the blocks of ONE stream with linked blocks (LZ4 frame format, lz4frame.h LZ4F_blockLinked), handled in order by one
 * wavefront.
 * Decompress, explicit batch: decoded into dst back to back (dst_off is ignored, total_bytes = capacity of dst), every block
 * may copy from what was decoded before it; src_len[i] bit 31 marks a stored block.
 * Compress, uniform batch (chunk_bytes = block size >= 64 KiB): block i is compressed with the hash table the blocks before it
 * left and may match up to 65535 bytes back into them (LZ4_compress_fast_continue in prefix mode, lz4.c:3179-3223); its output
 * goes to slot i with room for block size - 1 bytes (LZ4F_makeBlock, lz4frame.c:747-749); out_len[i] = 0: does not fit, the
 * caller stores the block.  Byte-identical to the blocks of the reference's LZ4F_compressFrame. */
#define AOCLGPU_FLAG_LZ4_LINKED 0x200u
/* Zstd compress: level 2 instead of level 1 (clevels.h:82,108: still the "fast" strategy for chunks <= 128 KiB, with a 2^15-slot
 * table - 68 KB of LDS, 2 chunks per CU - and minMatch 5 / 4); frames byte-identical to ZSTD_compress(.., 2). */
#define AOCLGPU_FLAG_ZSTD_LEVEL2 0x400u
/* Zstd compress: compression level 1..4 in bits 16-18 (0 there = level 1, or 2 with AOCLGPU_FLAG_ZSTD_LEVEL2).  Every chunk
 * gets the parameters the reference derives for (level, chunk size): levels 1-2 the "fast" strategy, levels 3-4 "double-fast"
 * (zstd_double_fast.c; two hash tables per resident workgroup in an HBM workspace; level 4 has no GPU parse for chunks of at
 * most 16 KiB, where the reference switches to "greedy": such chunks report 0).  AOCLGPU_FLAG_ZSTD_UPSTREAM_TABLE selects the
 * upstream level table (clevels.h:24-130) instead of the AOCL one (:155-262) the reference's dispatcher installs - they
 * differ in the double-fast rows only. */
#define AOCLGPU_ZSTD_LEVEL(l)            (((uint32_t)(l) & 7u) << 16)
#define AOCLGPU_FLAG_ZSTD_UPSTREAM_TABLE 0x80000u

/* Kernel selection.  None of these changes one output byte; they pin a choice the launcher otherwise makes from the
 * batch shape, so that tests can drive every kernel on small batches (tests/test_snappy_gpu.py, tests/test_lz4_gpu.py).
 *   Snappy compress: hash tables of all waves in LDS (snappy_compress_kernel), all in an HBM workspace
 *   (snappy_compress_gt_kernel), or the mixed-residency kernel (snappy_compress_mix_kernel).
 *   LZ4 compress: the u32 table with check bits (any chunk size) instead of the packed 18-bit table for chunks <= 256 KiB. */
#define AOCLGPU_FLAG_SNAPPY_TABLES_LDS   0x1000u
#define AOCLGPU_FLAG_SNAPPY_TABLES_HBM   0x2000u
#define AOCLGPU_FLAG_SNAPPY_TABLES_MIXED 0x4000u
#define AOCLGPU_FLAG_LZ4_TABLE_U32       0x0800u
/* LZ4 compress: the `acceleration` of LZ4_compress_fast (lz4.c:2937-2965; the probe step starts at `a` instead of 1 and
 * grows from there), 1 .. 65537, in bits 15..31 of the flags.  0 in those bits = acceleration 1 = LZ4_compress_default. */
#define AOCLGPU_LZ4_ACCELERATION(a)      ((uint32_t)(((a) < 1 ? 1 : ((a) > 65537 ? 65537 : (a))) - 1) << 15)

/* In uniform-mode DEcompression, src holds the slots (slot_bytes apart, src_len[i] = compressed size of
 * slot i, a device array that IS required), dst is the contiguous output: chunk i is written at
 * i*chunk_bytes and may hold up to min(chunk_bytes, total_bytes - i*chunk_bytes) bytes. */

int aoclgpu_device_count(void);
int aoclgpu_init(int device);                 /* selects the device, checks it is gfx950 */
const char* aoclgpu_last_error(void);
const char* aoclgpu_version(void);

/* worst-case compressed size of an n-byte chunk (lz4.h:261, snappy.cc:160-182, zstd.h:473) */
size_t aoclgpu_compress_bound(aoclgpu_codec codec, size_t n);

int aoclgpu_lz4_compress_chunks(const aoclgpu_batch* b, void* stream);
int aoclgpu_lz4_decompress_chunks(const aoclgpu_batch* b, void* stream);
/* The value LZ4_decompress_safe returns for a MALFORMED block of n bytes at src_dev (device memory) and an output capacity of
 * cap bytes: -(ip - src) - 1 with ip where the reference's decoder gives up (algos/lz4/lz4.c:3799-3802).  out_len < 0 of
 * aoclgpu_lz4_decompress_chunks only says that a chunk is malformed; this walks the block's tokens once more along the
 * reference's control flow (fast loop / safe loop, :3421-3790; nothing is decoded) and writes the value - or the decoded size,
 * should the block be well-formed - to *out (device-accessible memory).  lit_dist: 64 = the reference's AVX2 build of the
 * decoder (what it runs on an AVX2 host once set up), 32 = LZ4_decompress_generic.  The host one-shot call
 * (aoclgpu_decompress_host, LZ4_decompress_safe of include/aocl_native_api.h) does this by itself. */
int aoclgpu_lz4_error_value(const void* src_dev, uint32_t n, uint32_t cap, uint32_t lit_dist, int32_t* out, void* stream);
int aoclgpu_snappy_compress_chunks(const aoclgpu_batch* b, void* stream);
int aoclgpu_snappy_decompress_chunks(const aoclgpu_batch* b, void* stream);
int aoclgpu_zstd_compress_chunks(const aoclgpu_batch* b, void* stream);
int aoclgpu_zstd_decompress_chunks(const aoclgpu_batch* b, void* stream);
/* levels 3-4 on a large batch run as finder kernel + entropy kernel per slab of chunks (the slab size follows the device's free
 * memory): the number of slabs the newest such aoclgpu_zstd_compress_chunks call of this process used, 0 = one fused kernel */
int aoclgpu_zstd_last_split_slabs(void);

/* Zstd, one frame per input of ANY size below 2 GiB with the "fast" or "double-fast" strategy: what ZSTD_compress_advanced writes for these
 * compression parameters (zstd_compress.c:5388-5470 -> ZSTD_compress_frameChunk :4504-4583): blocks of min(128 KiB, window)
 * bytes that share the hash table, the repeat offsets and the last compressed block's Huffman table; RLE blocks after the
 * first; a window descriptor instead of the single-segment flag when the window is smaller than the input.  Blocks of a frame
 * depend on each other in order, so a frame is one wavefront's work; the frames of a batch run side by side.
 * Frame i covers src[src_off[i], src_off[i+1]) and may use dst[dst_off[i], dst_off[i+1]); out_len[i] = frame size, 0 when
 * it does not fit (ZSTD_error_dstSize_tooSmall).  All arrays are device memory, nframes + 1 offsets each. */
typedef struct {
    const uint8_t*  src;
    uint8_t*        dst;
    const uint64_t* src_off;
    const uint64_t* dst_off;
    int64_t*        out_len;
    uint32_t        nframes;
    uint32_t        windowLog, hashLog, minMatch;   /* ZSTD_compressionParameters */
    uint32_t        targetLength;                   /* fast: the parser's step is targetLength + !targetLength + 1 (zstd_fast.c:440): 2 for
                                                       0 and 1, 1 - level for the levels below -1 (at most 2^17); any value above 0
                                                       (row 0 of the level tables) leaves literals uncompressed
                                                       (zstd_compress_internal.h:609-622) */
    uint32_t        strategy;                       /* 1 = fast (zstd_fast.c), 2 = double-fast (zstd_double_fast.c: hashLog = long
                                                       table, chainLog = short table, targetLength unused) */
    uint32_t        chainLog;
    uint32_t        flags;                          /* 0, or AOCLGPU_ZSTD_FRAMES_FINDER_* : pins the choice the launcher otherwise makes
                                                       from the batch size (never changes an output byte; for tests and measurements) */
} aoclgpu_zstd_frames;
/* the fast strategy has two match finders: hash table in the HBM workspace, 9 frames resident per CU (the bulk variant), or hash
 * table and source ring in LDS with several sequences resolved per batch (hashLog <= 14, one frame per CU, ~2.4x faster per
 * frame: chosen for batches of at most two frames per CU) */
#define AOCLGPU_ZSTD_FRAMES_FINDER_HBM 1u
#define AOCLGPU_ZSTD_FRAMES_FINDER_LDS 2u
int aoclgpu_zstd_compress_frames(const aoclgpu_zstd_frames* f, void* stream);
/* the same for ONE input in host memory; checksumFlag appends the frame checksum (low 32 bits of XXH64 of the input).
 * Returns the frame size, 0 when it does not fit `cap`, < 0 on failure */
int64_t aoclgpu_zstd_compress_host_frame(const void* src_host, size_t n, void* dst_host, size_t cap,
                                         uint32_t strategy, uint32_t windowLog, uint32_t hashLog, uint32_t chainLog,
                                         uint32_t minMatch, uint32_t targetLength, uint32_t checksumFlag);

/* What the native Zstd entry points (ZSTD_compress, ZSTD_compressCCtx, ZSTD_compress_advanced, ZSTD_compress2, aocl_llc_compress)
 * do with an input above 128 KiB and "fast" parameters: 0 (default) = one frame of blocks, byte-identical to the reference,
 * serial per input (one wavefront, a few MiB/s); 1 = 128 KiB pieces as concatenated frames compressed side by side (any Zstd
 * decoder reads it as one stream, zstd_decompress.c:1113-1200; GiB/s; not the reference's bytes).  The environment variable
 * AOCLGPU_ZSTD_LARGE_INPUTS=pieces selects 1 at start-up. */
void aoclgpu_zstd_set_large_input_mode(int pieces);
int aoclgpu_zstd_large_input_mode(void);

/* LZ4F_compressFrame with LZ4F_blockLinked (the format's default) on an input of several blocks: 1 (default) = linked blocks,
 * byte-identical to the reference, one wavefront's sequential work (a few MiB/s); 0 = the blocks are written independent
 * instead (flagged so in the frame header; all blocks in one launch, GiB/s; every LZ4F decoder reads it; not the reference's
 * bytes).  AOCLGPU_LZ4F_LINKED=independent in the environment selects 0 at start-up. */
void aoclgpu_lz4f_set_block_linking(int linked);
int aoclgpu_lz4f_block_linking(void);

/* Compaction: gather the used prefix of every slot into one contiguous stream (the serial memcpy join
 * of the reference's MT path, lz4.c:2736-2906 / snappy.cc:2567-2654 / zstd_compress.c:5499-5517).
 * out_off[i] (device, uint64, nchunks+1 entries) receives the exclusive prefix sum of out_len. */
int aoclgpu_compact_slots(const uint8_t* slots, uint64_t slot_bytes, const int32_t* out_len, uint32_t nchunks,
                          uint8_t* packed, uint64_t* out_off, void* stream);

/* RAP ("random access point") streams, the reference's multi-threaded container (threads/threads.h:20-75):
 *   [ 'AOCL_LLC' | u32 frame length | u16 partitions | u16 0 | partitions x { u32 offset, u32 length, u32 decoded length } ]
 * followed by the partitions' payloads joined so that a legacy single-threaded decoder can decode the stream after
 * skipping the frame (aocl_skip_rap_frame_mt): LZ4 -> the final literal run of partition k is merged into the first
 * sequence of partition k+1 (lz4.c:2736-2906); Snappy -> one varint with the total length, partitions without their
 * own (snappy.cc:2567-2654); Zstd -> the frame sits in a skippable frame, partitions are whole frames
 * (zstd_compress.c:5343-5385,5499-5517).  With the reference's partition geometry the stream is byte-identical to
 * the reference's AOCL_ENABLE_THREADS build.
 *
 * aoclgpu_rap_pack: `compressed` is the batch that was just given to aoclgpu_<codec>_compress_chunks on the same
 * stream (uniform or explicit mode; LZ4 needs its aux[] output).  Writes the RAP stream to stream_out (device) and its
 * size to *stream_bytes (HOST).  One chunk -> no frame, the plain one-shot stream (as the reference with one thread).
 * The plan (12 bytes per partition) is made on the host: the call synchronises `stream`.
 * aoclgpu_rap_unpack: decodes a RAP stream (or a plain single-partition stream) produced by aoclgpu_rap_pack or by the
 * reference's MT build into dst (device); *decoded_bytes (HOST) = total size, negative = malformed.  Synchronous. */
int aoclgpu_rap_pack(aoclgpu_codec codec, const aoclgpu_batch* compressed, uint8_t* stream_out, uint64_t stream_cap,
                     uint64_t* stream_bytes, void* stream);
int aoclgpu_rap_unpack(aoclgpu_codec codec, const uint8_t* rap_stream, uint64_t stream_bytes, uint8_t* dst,
                       uint64_t dst_cap, int64_t* decoded_bytes, void* stream);
/* aoclgpu_rap_assemble: ONE RAP stream from chunks that were compressed elsewhere - on other GPUs - and gathered
 * (BASELINE config 5: the stream is sharded over the ranks by contiguous chunk ranges, every rank compacts its one-shot
 * chunk streams, the segments are exchanged).  `gathered` (device) holds chunk i's one-shot stream at chunk_off[i],
 * chunk_len[i] bytes; chunk_src_len[i] = its uncompressed size; chunk_aux[i] = the aux[] side output of its compress launch
 * (LZ4 only, may be NULL otherwise).  The four arrays are HOST arrays in stream order (rank 0's chunks, rank 1's, ...).
 * Nothing but the compressed chunks is needed: what LZ4 carries across a partition boundary (lz4.c:2823-2901) are the last
 * chunk_aux[i] bytes of chunk i's stream.  Same output as aoclgpu_rap_pack over one device's batch of the same chunks; at
 * most 65535 chunks and 2 GiB of source per stream (16-bit / 32-bit RAP fields): longer streams are assembled slab by slab. */
int aoclgpu_rap_assemble(aoclgpu_codec codec, const uint8_t* gathered, const uint64_t* chunk_off, const int32_t* chunk_len,
                         const uint32_t* chunk_src_len, const uint32_t* chunk_aux, uint32_t nchunks,
                         uint8_t* stream_out, uint64_t stream_cap, uint64_t* stream_bytes, void* stream);
/* bytes of RAP framing aoclgpu_rap_pack adds for nchunks partitions (0 for one chunk) */
uint64_t aoclgpu_rap_frame_bytes(aoclgpu_codec codec, uint32_t nchunks);
/* Host-buffer forms (H2D, chunked compress + pack / unpack + chunked decompress, D2H): what the reference's
 * AOCL_ENABLE_THREADS build does inside its one-shot calls, with GPU chunks as the partitions.  chunk_bytes 0 = the
 * codec's default partition (aoclgpu_rap_default_chunk: LZ4 256 KiB, Snappy 64 KiB, Zstd 128 KiB).  Inputs up to 2 GiB.
 * Return conventions of aoclgpu_compress_host / aoclgpu_decompress_host. */
uint64_t aoclgpu_rap_default_chunk(aoclgpu_codec codec);
int64_t aoclgpu_rap_compress_host(aoclgpu_codec codec, const void* host_src, size_t n, void* host_dst, size_t cap,
                                  uint64_t chunk_bytes);
int64_t aoclgpu_rap_decompress_host(aoclgpu_codec codec, const void* host_src, size_t n, void* host_dst, size_t cap);
/* the same with the codec's per-launch flags for every partition (AOCLGPU_ZSTD_LEVEL(2..4), AOCLGPU_LZ4_ACCELERATION(a), ...) */
int64_t aoclgpu_rap_compress_host_ex(aoclgpu_codec codec, const void* host_src, size_t n, void* host_dst, size_t cap,
                                     uint64_t chunk_bytes, uint32_t flags);
/* Zstd RAP stream in the REFERENCE'S partition geometry (host buffers): the bytes ZSTD_compress_advanced of the reference's
 * AOCL_ENABLE_THREADS build writes when `nparts` threads take part (zstd_compress.c:5410-5522, threads/threads.c:46-136):
 * nparts partitions of n / nparts bytes, the last one with the leftover, each ONE frame with the given parameters (one
 * wavefront per partition), joined behind the RAP frame.  Correct for every fast / double-fast parameter set, slow by
 * construction (few, large partitions): bulk callers use aoclgpu_rap_compress_host_ex (128 KiB partitions).
 * Returns the stream size, 0 when it does not fit `cap`, <= AOCLGPU_HOST_FAIL / 2 on a GPU failure. */
int64_t aoclgpu_zstd_rap_compress_host_parts(const void* host_src, size_t n, void* host_dst, size_t cap, uint32_t nparts,
                                             uint32_t strategy, uint32_t windowLog, uint32_t hashLog, uint32_t chainLog,
                                             uint32_t minMatch, uint32_t targetLength);
/* aoclgpu_rap_assemble_slabs: a stream of ANY length as the sequence of RAP streams the container allows (32-bit offsets,
 * 16-bit partition count: threads/threads.h:77-106; the reference's bench cuts its input into slabs the same way,
 * test/codec_bench.h:58).  Consecutive chunks are grouped into slabs of at most slab_src_bytes source bytes (0 = 1 GiB, at
 * most 2 GiB - 1) and 65535 partitions; every slab is one aoclgpu_rap_assemble stream, written back to back into stream_out.
 * slab_off[0 .. *nslabs] = their offsets in stream_out, slab_first[0 .. *nslabs] = the first chunk of each (host arrays with
 * room for max_slabs + 1 entries).  The per-chunk arrays are host arrays as for aoclgpu_rap_assemble. */
/* ---- BASELINE config 5 from C: the stream sharded over the GPUs of a node, compressed segments exchanged over RCCL / xGMI ----
 * One process (or thread) per GPU.  Rank 0 makes an id (aoclgpu_comm_unique_id) and hands it to the other ranks by whatever
 * the host program uses (MPI, a file, torch.distributed ...); every rank calls aoclgpu_comm_create on ITS current device.
 * `comm` is an ncclComm_t: one created by the caller with ncclCommInitRank works as well.  RCCL (librccl.so.1) is bound at
 * run time on first use; without it these calls fail with AOCLGPU_ERR_HIP and the rest of the library is unaffected.
 *   per step and rank:  aoclgpu_<codec>_compress_chunks  ->  aoclgpu_compact_slots straight into gather_buf + seg_off[rank]
 *                       ->  aoclgpu_allgather_words (per-chunk sizes; LZ4 also the aux[] run lengths)
 *                       ->  aoclgpu_exchange_segments  ->  aoclgpu_rap_assemble_slabs on the rank(s) that write the stream.
 * aoclgpu_exchange_segments: afterwards gather_buf[seg_off[r], +seg_bytes[r]) holds rank r's segment on every rank (this
 * rank's own must already be there); one group of exact-size ncclSend / ncclRecv pairs on `stream`, every pair of ranks over
 * its own xGMI link.  seg_bytes / seg_off: host arrays of nranks entries, the same on all ranks.
 * aoclgpu_allgather_words: all_dev[r * nwords ..] = rank r's mine_dev[0 .. nwords) (32-bit words, ncclAllGather).
 * aoclgpu_comm_count: *nranks = the number of ranks RCCL itself holds for `comm` (ncclCommCount) - what a bench line records so
 * that a run whose communicator came up smaller than the launch says so. */
#define AOCLGPU_COMM_ID_BYTES 128
int aoclgpu_comm_unique_id(uint8_t id[AOCLGPU_COMM_ID_BYTES]);
int aoclgpu_comm_create(const uint8_t id[AOCLGPU_COMM_ID_BYTES], int nranks, int rank, void** comm);
int aoclgpu_comm_destroy(void* comm);
int aoclgpu_comm_count(void* comm, int* nranks);
int aoclgpu_exchange_segments(void* comm, void* gather_buf, const uint64_t* seg_bytes, const uint64_t* seg_off,
                              int nranks, int rank, void* stream);
int aoclgpu_allgather_words(void* comm, const void* mine_dev, void* all_dev, size_t nwords, int nranks, int rank, void* stream);

/* aocl_llc_compress(ZSTD) with numThreads > 1: 0 (default) = 128 KiB partitions (the GPU's geometry), 1 = the reference's
 * geometry (aoclgpu_zstd_rap_compress_host_parts with the partition count threads/threads.c computes for numThreads);
 * the environment variable AOCLGPU_RAP_GEOMETRY=reference selects 1 at start-up. */
void aoclgpu_set_rap_geometry(int reference);
int aoclgpu_rap_geometry(void);
int aoclgpu_rap_assemble_slabs(aoclgpu_codec codec, const uint8_t* gathered, const uint64_t* chunk_off, const int32_t* chunk_len,
                               const uint32_t* chunk_src_len, const uint32_t* chunk_aux, uint64_t nchunks, uint64_t slab_src_bytes,
                               uint8_t* stream_out, uint64_t stream_cap, uint64_t* slab_off, uint64_t* slab_first, uint32_t max_slabs,
                               uint32_t* nslabs, void* stream);

/* Timing helper used by bench.py: runs `iters` back-to-back launches of one batch op on `stream`
 * bracketed by HIP events recorded on that same stream; returns the average milliseconds per launch
 * (negative on error).  op: 0 = compress, 1 = decompress. */
double aoclgpu_time_batch(aoclgpu_codec codec, int op, const aoclgpu_batch* b, int iters, void* stream);

/* Memory-system probes for bench.py's roofline ("a device-to-device copy kernel gives the achievable ceiling").
 * aoclgpu_probe_copy: average milliseconds of one device-to-device copy of `bytes` (16-byte lanes, coalesced) - of the fastest
 * of a few loop shapes (tile depth, non-temporal or not, workgroups per CU; aoclgpu_probe_copy_variant names the winner); the
 * achievable HBM rate is 2 * bytes / time.  aoclgpu_probe_scatter8: one launch of `loads_per_lane` independent 8-byte loads per lane at
 * pseudo-random addresses of src[0, bytes) - the codecs' candidate-window pattern; returns its milliseconds, *total_loads =
 * loads issued (under rocprofv3 --pmc FETCH_SIZE this calibrates the counter for scattered loads).  Negative on error. */
double aoclgpu_probe_copy(void* dst_dev, const void* src_dev, size_t bytes, int iters, void* stream);
const char* aoclgpu_probe_copy_variant(void);
double aoclgpu_probe_scatter8(const void* src_dev, size_t bytes, uint32_t loads_per_lane, uint64_t* total_loads, void* stream);

/* Host-buffer convenience (H2D, one launch of ONE chunk, D2H).  Return value = the per-chunk result of the
 * native one-shot API (compress: size, 0 = does not fit; decompress: size, small negative = malformed),
 * or a value <= AOCLGPU_HOST_FAIL/2 when the GPU path itself failed (AOCLGPU_HOST_FAIL + AOCLGPU_ERR_*). */
#define AOCLGPU_HOST_FAIL (-(1LL << 40))
/* flags: AOCLGPU_FLAG_* of the codec (e.g. AOCLGPU_FLAG_ZSTD_LEVEL2) */
/* The Zstd match finder alone ("fast" parse of level 1, no history): sequences of ONE chunk (7 .. 131072 bytes, device memory).
 * out_host[0] = number of sequences, out_host[1] = total literal bytes, then {litLength, offBase, matchLength} per sequence
 * (offBase as in zstd: 1 = repcode, offset + 3 otherwise).  Synchronous; flags: 0. */
int aoclgpu_zstd_block_sequences(const void* src_dev, uint32_t n, uint32_t* out_host, uint32_t max_sequences, uint32_t flags);

/* Zstd, inputs of any size on host buffers: the input is cut into 128 KiB pieces, each becomes an independent frame
 * (one launch for all of them), the frames are concatenated.  flags: AOCLGPU_FLAG_ZSTD_LEVEL2.  Returns the stream size,
 * 0 when it does not fit `cap`, or <= AOCLGPU_HOST_FAIL/2 on a GPU failure.
 * aoclgpu_zstd_decompress_host_frames: any Zstandard stream; a stream of several frames that all carry their content
 * size is split on the host (frame and block headers only) and its frames are decoded in parallel, one wavefront each,
 * otherwise one wavefront walks the stream.  Returns the decoded size, or -(the error code ZSTD_decompress returns for the
 * stream, zstd_errors.h:60-106: 20 corruption_detected, 70 dstSize_tooSmall, 72 srcSize_wrong, 10 prefix_unknown, ...) -
 * the code of the reference's check that fails first, frame by frame and block by block (INTEGRATION.md lists where the decoder
 * is stricter than the reference's). */
int64_t aoclgpu_zstd_compress_host_frames(const void* host_src, size_t n, void* host_dst, size_t cap, uint32_t flags);
int64_t aoclgpu_zstd_decompress_host_frames(const void* host_src, size_t n, void* host_dst, size_t cap);

/* Which GPU the host-buffer entry points (this header's *_host* functions, aocl_llc_*, the native names) use: default 0,
 * or the environment variable AOCLGPU_DEVICE read at first use. */
int aoclgpu_set_host_device(int device);
int aoclgpu_host_device(void);
/* The host-buffer entry points keep their device staging buffers between calls (up to 256 MiB each; larger ones are freed when
 * the call returns); the RAP host calls (aoclgpu_rap_compress_host*, aocl_llc_compress with numThreads > 1) keep ONE buffer set
 * per process - ~3.3x the input - while it is at most AOCLGPU_HOST_CACHE_KEEP_MB (default 4096: one <= 1 GiB call, the slab
 * of the reference's bench; a call that needed more frees everything on return).  aoclgpu_host_cache_bytes() = what that
 * cache holds now.  The launchers keep one scratch allocation per (device, stream) - hash tables and sequence stores of
 * the resident chunks; after a Zstd level 3-4 batch of thousands of chunks up to AOCLGPU_ZSTD_DF_SLAB_MB (default 32 GiB) of
 * per-chunk workspaces.  aoclgpu_host_trim frees the staging buffers of every idle context and all launcher scratch (it waits
 * for the work that still uses it); aocl_llc_destroy calls it. */
void aoclgpu_host_trim(void);
uint64_t aoclgpu_host_cache_bytes(void);

int64_t aoclgpu_compress_host_ex(aoclgpu_codec codec, const void* host_src, size_t n, void* host_dst, size_t cap, uint32_t flags);
int64_t aoclgpu_compress_host(aoclgpu_codec codec, const void* host_src, size_t n, void* host_dst, size_t cap);
int64_t aoclgpu_decompress_host(aoclgpu_codec codec, const void* host_src, size_t n, void* host_dst, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
