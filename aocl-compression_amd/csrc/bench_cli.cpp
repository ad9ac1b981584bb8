// csrc/bench_cli.cpp -- `aocl_compression_bench` for the GPU build: the counterpart of the reference's test bench
// (test/codec_bench.c:86-126 options, :306-497 option parsing, :773-1133 run / verify / statistics), same flags and
// the same printed lines, so the reference's CTest-style corpus sweeps (test/CMakeLists.txt:77-105) can be pointed
// at libaocl_compression_gpu.so.
//
//   aocl_compression_bench -e<method>[:<level>[:<opt>]] [-i<iters>] [-t] [-p] [-m<MB>] [-r<compress|decompress>]
//                          [-d<dump file>] [-f<validation file>] [-n] [-o] [-a] [-l] [-h] <input file>
//
// The file is processed in slabs of at most min(file size, -m MB, 1024 MB) like the reference (:516-536,:806-826).
// Unified-API mode drives aocl_llc_compress / aocl_llc_decompress with numThreads = 2, i.e. every slab becomes one
// RAP stream of GPU partitions (what the reference bench gets from an AOCL_ENABLE_THREADS build); -n uses the native
// one-shot entry points (LZ4_compress_default ... ZSTD_decompress): ONE chunk = one wavefront per slab, bit-exact
// to the single-threaded reference and only sensible for small inputs.  Methods without a GPU backend are listed
// by -l as the reference lists them and fail with the reference's "not supported" status.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>
#include <algorithm>
#include <string>
#include <vector>
#include "../../include/aocl_compression.h"
#include "../../include/aocl_gpu_codec.h"
#include "../../include/aocl_native_api.h"

namespace {

struct Method { const char* name; aocl_compression_type type; long lower, upper, def; bool gpu; };
const Method kMethods[] = {                       // codec_list, test/codec_bench.h
    {"LZ4", LZ4, 0, 0, 0, true}, {"LZ4HC", LZ4HC, 1, 12, 9, false}, {"LZMA", LZMA, 0, 9, 1, false},
    {"BZIP2", BZIP2, 1, 9, 6, false}, {"SNAPPY", SNAPPY, 0, 0, 0, true}, {"ZLIB", ZLIB, 1, 9, 6, false},
    {"ZSTD", ZSTD, 1, 22, 1, true},        // default level of the GPU build: 1 (GPU backend: levels 1-4 and the negative levels)
};
enum { RUN_DEFAULT, RUN_COMPRESS, RUN_DECOMPRESS };

struct Opts {
    int method = -1; long level = -1000; long optVar = 0; int iters = 10; bool verify = false, stats = false, native = false, all = false;
    size_t memMB = 1024; int run = RUN_DEFAULT; std::string dump, valFile, input;
};

uint64_t now_ns() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (uint64_t)ts.tv_sec * 1000000000ull + ts.tv_nsec; }

void usage()
{
    printf("\nAOCL Compression Library version: %s\n", aocl_llc_version());
    printf("Usage: aocl_compression_bench <options> input\n\n");
    printf("where input is the test file name and <options> can be:\n");
    printf("-h | --help Print help info\n");
    printf("-l          List all the available compression/decompression methods\n");
    printf("-a          Use all the available compression/decompression methods\n");
    printf("-m<>        Maximum size in MBs of the input for compression and decompression. Default=MIN(filesize, 1024 MB)\n");
    printf("-e<>:<>:<>  Compression/decompression method. Optional level and additional param may be specified using : separator\n");
    printf("-i<>        Number of iterations of compression/decompression\n");
    printf("-t          Verification and functional tests of the compression/decompression methods\n");
    printf("-p          Print stats like compression/decompression time, speed, ratio\n");
    printf("-o          Turn off all optimizations (accepted; the GPU build has no host ISA paths to turn off)\n");
    printf("-r          Run particular operation only. Supported options are 'compress' and 'decompress'.\n\n");
    printf("-d          File to dump output data. Based on -r, saves compressed/decompressed data.\n\n");
    printf("-f          Input uncompressed file to be used for validation in -rdecompress mode.\n\n");
    printf("-n          Use Native APIs for compression/decompression.\n\n");
}

void list_methods()
{
    printf("\nSupported compression/decompression methods along with their supported levels are:\n\n");
    printf("===========================================\nMethod Name\tLower Level\tUpper Level\n===========================================\n");
    for (const Method& m : kMethods) {
        if (m.lower == m.upper) printf("%s\t\t %s\t\t%s%s\n", m.name, "NA", "NA", m.gpu ? "" : "\t(no GPU backend)");
        else printf("%s\t\t %ld\t\t%ld%s\n", m.name, m.lower, m.upper, m.gpu ? (m.type == ZSTD ? "\t(GPU backend: levels 1-4 and negative levels)" : "") : "\t(no GPU backend)");
    }
    printf("\n");
}

bool read_file(const std::string& path, std::vector<char>& out)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    const bool ok = n <= 0 || fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

int parse(int argc, char** argv, Opts& o)
{
    if (argc <= 1) { usage(); return 2; }
    for (int i = 1; i < argc; ++i) {
        const char* a = argv[i];
        if (a[0] != '-') { if (!o.input.empty()) { printf("Invalid option passed, ignoring more than one input file. Use -h to know supported user options.\n"); } else o.input = a; continue; }
        switch (a[1]) {
        case 'h': usage(); return 2;
        case '-': if (!strcmp(a, "--help")) { usage(); return 2; } printf("Invalid option passed. Use -h to know supported user options.\n"); return -1;
        case 'l': list_methods(); return 2;
        case 'a': o.all = true; break;
        case 'm': o.memMB = (size_t)atol(a + 2); if (o.memMB == 0) o.memMB = 1024; break;
        case 'e': {
            std::string s = a + 2; std::string name = s.substr(0, s.find(':'));
            for (char& c : name) c = (char)toupper(c);
            for (size_t k = 0; k < sizeof(kMethods) / sizeof(kMethods[0]); ++k) if (name == kMethods[k].name) o.method = (int)k;
            if (o.method < 0) { printf("Unsupported option passed. Use -l to know supported methods and -h for exact user options.\n"); return -1; }
            size_t c1 = s.find(':');
            if (c1 != std::string::npos) { o.level = atol(s.c_str() + c1 + 1); size_t c2 = s.find(':', c1 + 1); if (c2 != std::string::npos) o.optVar = atol(s.c_str() + c2 + 1); }
            break;
        }
        case 'i': o.iters = atoi(a + 2); if (o.iters < 1) o.iters = 1; break;
        case 't': o.verify = true; break;
        case 'p': o.stats = true; break;
        case 'n': o.native = true; break;
        case 'o': break;
        case 'c': printf("IPP methods are not part of the GPU build.\n"); return -1;
        case 'r':
            if (o.run != RUN_DEFAULT) { printf("Multiple -r options are not allowed.\n\n"); return -1; }
            if (!strcmp(a + 2, "compress")) o.run = RUN_COMPRESS; else if (!strcmp(a + 2, "decompress")) o.run = RUN_DECOMPRESS;
            else { printf("Invalid -r option. Valid options are 'compress' and 'decompress'\n\n"); return -1; }
            break;
        case 'd': if (!o.dump.empty()) { printf("Multiple -d options are not allowed.\n\n"); return -1; } o.dump = a + 2; break;
        case 'f': if (!o.valFile.empty()) { printf("Multiple -f options are not allowed.\n\n"); return -1; } o.valFile = a + 2; break;
        default: printf("Invalid option passed. Use -h to know supported user options.\n"); return -1;
        }
    }
    if (o.input.empty()) { printf("Invalid option passed. Use -h to know supported user options.\n"); return -1; }
    if (o.run == RUN_DECOMPRESS && o.method < 0) { printf("Codec not specified. Specify codec using -e when running -rdecompress mode.\n\n"); return -1; }
    return 0;
}

int64_t native_compress(aocl_compression_type t, const char* in, size_t n, char* out, size_t cap, long level)
{
    if (t == LZ4) { int r = LZ4_compress_default(in, out, (int)n, (int)(cap > 0x7FFFFFFF ? 0x7FFFFFFF : cap)); return r > 0 ? r : -1; }
    if (t == SNAPPY) { size_t l = cap; return snappy_compress(in, n, out, &l) == SNAPPY_OK ? (int64_t)l : -1; }
    size_t r = ZSTD_compress(out, cap, in, n, (int)level);
    return ZSTD_isError(r) ? -1 : (int64_t)r;
}
int64_t native_decompress(aocl_compression_type t, const char* in, size_t n, char* out, size_t cap)
{
    if (t == LZ4) { int r = LZ4_decompress_safe(in, out, (int)n, (int)(cap > 0x7FFFFFFF ? 0x7FFFFFFF : cap)); return r >= 0 ? r : -1; }
    if (t == SNAPPY) { size_t l = cap; return snappy_uncompress(in, n, out, &l) == SNAPPY_OK ? (int64_t)l : -1; }
    size_t r = ZSTD_decompress(out, cap, in, n);
    return ZSTD_isError(r) ? -1 : (int64_t)r;
}

int run_method(const Opts& o, const Method& m, const std::vector<char>& file, const std::vector<char>& val)
{
    const long level = (m.lower == m.upper) ? 0 : (o.level == -1000 ? m.def : o.level);
    const char* fname = o.input.c_str();
    if (!m.gpu) { printf("AOCL-COMPRESSION [%s-%ld] [Filename:%s] Compression: failed (method has no GPU backend)\n", m.name, level, fname); return ERR_EXCLUDED_METHOD; }
    aocl_compression_desc h; memset(&h, 0, sizeof(h));
    h.level = (size_t)level; h.optVar = (size_t)o.optVar; h.numThreads = 2; h.measureStats = 0; h.optOff = 0;
    if (aocl_llc_setup(&h, m.type) != 0) { printf("AOCL-COMPRESSION [%s-%ld] [Filename:%s] setup failed: %s\n", m.name, level, fname, aoclgpu_last_error()); return -1; }
    const size_t slab = std::min(file.size() ? file.size() : (size_t)1, std::min(o.memMB, (size_t)1024) << 20);
    const size_t outCap = aoclgpu_compress_bound(m.type == LZ4 ? AOCLGPU_LZ4 : (m.type == SNAPPY ? AOCLGPU_SNAPPY : AOCLGPU_ZSTD), slab) + slab / 6 +
                          (size_t)aocl_get_rap_frame_bound_mt() + 64;
    std::vector<char> comp(outCap), back(o.run == RUN_DECOMPRESS ? 0 : slab);
    uint64_t cTime = 0, dTime = 0, cBest = ~0ull, dBest = ~0ull, cSize = 0, dSize = 0;
    int status = 0;
    FILE* dumpFp = o.dump.empty() ? nullptr : fopen(o.dump.c_str(), "wb");
    if (!o.dump.empty() && !dumpFp) { printf("Error in opening output file [%s].\n", o.dump.c_str()); aocl_llc_destroy(&h, m.type); return -1; }
    if (o.run != RUN_DECOMPRESS) {
        for (int it = 0; it < o.iters && !status; ++it) {
            uint64_t ct = 0, dt = 0;
            for (size_t pos = 0; pos < std::max(file.size(), (size_t)1) && !status; pos += slab) {
                const size_t n = std::min(slab, file.size() - pos);
                uint64_t t0 = now_ns();
                int64_t r;
                if (o.native) r = native_compress(m.type, file.data() + pos, n, comp.data(), outCap, level);
                else { h.inBuf = (char*)file.data() + pos; h.inSize = n; h.outBuf = comp.data(); h.outSize = outCap; r = aocl_llc_compress(&h, m.type); }
                ct += now_ns() - t0;
                if (r <= 0) { printf("AOCL-COMPRESSION [%s-%ld] [Filename:%s] Compression: failed\n", m.name, level, fname); status = -1; break; }
                cSize += (uint64_t)r;
                if (dumpFp && it == 0 && o.run == RUN_COMPRESS && fwrite(comp.data(), 1, (size_t)r, dumpFp) != (size_t)r) { printf("AOCL-COMPRESSION [%s-%ld] [Filename:%s] Dump: failed\n", m.name, level, fname); status = -1; break; }
                if (o.run == RUN_COMPRESS) continue;
                t0 = now_ns();
                int64_t d;
                if (o.native) d = native_decompress(m.type, comp.data(), (size_t)r, back.data(), n);
                else { h.inBuf = comp.data(); h.inSize = (size_t)r; h.outBuf = back.data(); h.outSize = n; d = aocl_llc_decompress(&h, m.type); }
                dt += now_ns() - t0;
                if (d < 0) { printf("AOCL-COMPRESSION [%s-%ld] [Filename:%s] Decompression: failed\n", m.name, level, fname); status = -1; break; }
                dSize += (uint64_t)d;
                if (o.verify && ((size_t)d != n || memcmp(back.data(), file.data() + pos, n) != 0)) { printf("AOCL-COMPRESSION [%s-%ld] [Filename:%s] verification: failed\n", m.name, level, fname); status = -1; break; }
                if (file.empty()) break;
            }
            cTime += ct; dTime += dt; if (ct < cBest) cBest = ct; if (dt < dBest) dBest = dt;
        }
    } else {                                                      // -rdecompress: the input file is one compressed stream
        const size_t cap = val.empty() ? std::max(file.size() * 64, (size_t)1 << 20) : val.size();
        back.resize(cap);
        for (int it = 0; it < o.iters && !status; ++it) {
            uint64_t t0 = now_ns();
            int64_t d;
            if (o.native) d = native_decompress(m.type, file.data(), file.size(), back.data(), cap);
            else { h.inBuf = (char*)file.data(); h.inSize = file.size(); h.outBuf = back.data(); h.outSize = cap; d = aocl_llc_decompress(&h, m.type); }
            const uint64_t dt = now_ns() - t0;
            if (d < 0) { printf("AOCL-COMPRESSION [%s-%ld] [Filename:%s] Decompression: failed\n", m.name, level, fname); status = -1; break; }
            dSize += (uint64_t)d; dTime += dt; if (dt < dBest) dBest = dt;
            if (dumpFp && it == 0 && fwrite(back.data(), 1, (size_t)d, dumpFp) != (size_t)d) { printf("AOCL-COMPRESSION [%s-%ld] [Filename:%s] Dump: failed\n", m.name, level, fname); status = -1; break; }
            if (o.verify) {
                if (val.empty()) { printf("AOCL-COMPRESSION [%s-%ld] [Filename:%s] verification file not provided\n", m.name, level, fname); status = -1; break; }
                if ((size_t)d != val.size() || memcmp(back.data(), val.data(), val.size()) != 0) { printf("AOCL-COMPRESSION [%s-%ld] [Filename:%s] verification: failed\n", m.name, level, fname); status = -1; break; }
            }
        }
    }
    if (dumpFp) fclose(dumpFp);
    aocl_llc_destroy(&h, m.type);
    if (status) return status;
    if (o.run != RUN_COMPRESS && o.verify) printf("AOCL-COMPRESSION [%s-%ld] [Filename:%s] verification: passed\n", m.name, level, fname);
    if (o.stats) {                                                // test/codec_bench.c:1069-1128, same arithmetic
        printf("AOCL-COMPRESSION [%s-%ld] [Filename:%s] -------------------------------------\n", m.name, level, fname);
        const double fsz = (double)file.size();
        if (o.run != RUN_DECOMPRESS)
            printf("Compression:         speed(avg) %.2f MB/s, time(avg) %.2f ms, size %zu, speed(best) %.2f MB/s, time(best) %.2f ms\n",
                   fsz * o.iters * 1000.0 / (double)cTime, cTime / (o.iters * 1000000.0), (size_t)(cSize / o.iters), fsz * 1000.0 / (double)cBest, cBest / 1000000.0);
        if (o.run != RUN_COMPRESS) {
            const double dsz = o.run == RUN_DEFAULT ? fsz : (double)(dSize / o.iters);
            printf("Decompression:       speed(avg) %.2f MB/s, time(avg) %.2f ms, size %zu, speed(best) %.2f MB/s, time(best) %.2f ms\n",
                   dsz * o.iters * 1000.0 / (double)dTime, dTime / (o.iters * 1000000.0), (size_t)(dSize / o.iters), dsz * 1000.0 / (double)dBest, dBest / 1000000.0);
        }
        if (o.run != RUN_DECOMPRESS) printf("Ratio:               %.2f\n", (cSize * 100.0 / o.iters) / fsz);
    }
    return 0;
}

}  // namespace

int main(int argc, char** argv)
{
    Opts o;
    const int pr = parse(argc, argv, o);
    if (pr == 2) return 0;
    if (pr) return 1;
    std::vector<char> file, val;
    if (!read_file(o.input, file)) { printf("Error in opening input file [%s].\n", o.input.c_str()); return 1; }
    if (!o.valFile.empty() && !read_file(o.valFile, val)) { printf("Error in opening input file [%s].\n", o.valFile.c_str()); return 1; }
    int rc = 0;
    if (o.all || o.method < 0) {
        for (const Method& m : kMethods) if (m.gpu) rc |= run_method(o, m, file, val) != 0;
    } else rc = run_method(o, kMethods[o.method], file, val) != 0;
    return rc ? 1 : 0;
}
