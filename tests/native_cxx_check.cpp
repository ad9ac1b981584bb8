// A C++ caller written against the reference's snappy.h / zstd.h names, linked against libaocl_compression_gpu.so
// (tests/test_native_api.py builds and runs it on the GPU box): the std::string flavour of the Snappy API and the
// context-parameter flavour of the Zstd API.
#include "../include/aocl_native_api.h"
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

int main()
{
    std::string text;
    for (int i = 0; i < 20000; ++i) text += "the quick brown fox " + std::to_string(i % 97) + " jumps over the lazy dog; ";
    // snappy::Compress / Uncompress (snappy.h:157-180)
    std::string c, back;
    const size_t n = snappy::Compress(text.data(), text.size(), &c);
    if (n == 0 || n != c.size() || n >= text.size()) { std::printf("snappy::Compress failed (%zu)\n", n); return 1; }
    if (!snappy::IsValidCompressedBuffer(c.data(), c.size())) { std::printf("not valid\n"); return 1; }
    if (!snappy::Uncompress(c.data(), c.size(), &back) || back != text) { std::printf("snappy::Uncompress failed\n"); return 1; }
    std::string empty_c, empty_back = "x";
    if (snappy::Compress("", 0, &empty_c) != 1 || !snappy::Uncompress(empty_c.data(), empty_c.size(), &empty_back) || !empty_back.empty()) { std::printf("empty input\n"); return 1; }
    if (snappy::Uncompress("\xff\xff\xff", 3, &back)) { std::printf("garbage accepted\n"); return 1; }
    // ZSTD_compress2 with the level stored in the context (zstd.h:1057), ZSTD_CCtx_reset, frame inspection
    ZSTD_CCtx* cctx = ZSTD_createCCtx();
    std::vector<char> z(ZSTD_compressBound(text.size())), d(text.size());
    size_t sizes[5] = {0, 0, 0, 0, 0};
    for (int level = 1; level <= 4; ++level) {
        if (ZSTD_isError(ZSTD_CCtx_setParameter(cctx, ZSTD_c_compressionLevel, level))) { std::printf("setParameter\n"); return 1; }
        const size_t r = ZSTD_compress2(cctx, z.data(), z.size(), text.data(), text.size());
        if (ZSTD_isError(r)) { std::printf("ZSTD_compress2 level %d: %s\n", level, ZSTD_getErrorName(r)); return 1; }
        const size_t one = ZSTD_compress(d.data(), d.size(), text.data(), text.size(), level);
        if (one != r || std::memcmp(d.data(), z.data(), r) != 0) { std::printf("compress2 != compress at level %d\n", level); return 1; }
        if (ZSTD_decompressBound(z.data(), r) != text.size()) { std::printf("decompressBound\n"); return 1; }
        size_t pos = 0, frames = 0;
        while (pos < r) { const size_t f = ZSTD_findFrameCompressedSize(z.data() + pos, r - pos); if (ZSTD_isError(f)) { std::printf("findFrameCompressedSize\n"); return 1; } pos += f; frames++; }
        if (frames != 1) { std::printf("frames %zu at level %d\n", frames, level); return 1; }   // one frame of blocks, like the reference
        const size_t dr = ZSTD_decompress(d.data(), d.size(), z.data(), r);
        if (dr != text.size() || std::memcmp(d.data(), text.data(), dr) != 0) { std::printf("round trip level %d\n", level); return 1; }
        sizes[level] = r;
    }
    ZSTD_CCtx_reset(cctx, ZSTD_reset_session_and_parameters);         // back to the default level (3)
    if (ZSTD_compress2(cctx, z.data(), z.size(), text.data(), text.size()) != sizes[3]) { std::printf("reset\n"); return 1; }
    ZSTD_CCtx_setParameter(cctx, ZSTD_c_compressionLevel, 12);
    if (!ZSTD_isError(ZSTD_compress2(cctx, z.data(), z.size(), text.data(), text.size()))) { std::printf("level 12 accepted\n"); return 1; }
    ZSTD_freeCCtx(cctx);
    std::printf("native C++ check OK (snappy %zu -> %zu, zstd levels 1-4: %zu %zu %zu %zu)\n", text.size(), n, sizes[1], sizes[2], sizes[3], sizes[4]);
    return 0;
}
