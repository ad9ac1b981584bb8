/* tools/experiments/lz4_segment_study.c -- CPU study for verdict r04 item 7 (single-buffer throughput without giving up the bytes).
 *
 * Question: a byU32 LZ4 parser (lz4.c:1371-1845 as restated in oracle/lz4_oracle.c; candidates further than 65535 bytes are
 * rejected, :1556-1564) that is started COLD `warm` bytes in front of a segment boundary B - empty table, anchor = ip - 1 -
 * how often, and how soon behind B, is it in the state of the true parser (the one that started at byte 0)?
 *
 * For every boundary the study runs the cold parser from B - warm to B + zone and reports
 *   sync      first position p >= B at which both parsers stand at the top of their search loop (same ip, anchor = ip - 1)
 *   strict    ... and the two hash tables agree at p on every entry that can still matter (position >= p - 65535): from p on
 *             the cold parser's output IS the reference's - provable from the two states alone
 *   actual    ground truth: from the first such p on, both parsers emit the same sequences up to the end of the zone
 * The parser below follows oracle/lz4_oracle.c (byU32 branch) statement for statement, with hooks.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

enum { MINMATCH = 4, LASTLITERALS = 5, MFLIMIT = 12, DIST_MAX = 65535, SKIPTRIGGER = 6, HASHLOG = 12 };
static uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static uint32_t hash5(const uint8_t* p) { return (uint32_t)(((rd64(p) << 24) * 889523592379ULL) >> (64 - HASHLOG)); }

typedef struct { uint32_t start, lit, off, len; } Seq;       /* literals [start, start + lit), match of len bytes at distance off */
typedef struct {
    const uint8_t* src; uint32_t n;
    uint32_t table[1 << HASHLOG];
    uint32_t ip, anchor;
    uint32_t floor;              /* cold start: positions below `floor` do not exist for this parser (an empty slot holds floor) */
    uint8_t* ss;                 /* bitmap over positions: top of the search loop reached with this ip */
    uint8_t* wr;                 /* bitmap over positions: a table write of this position happened */
    Seq* seqs; size_t nseq, cap;
} Parser;

static void setbit(uint8_t* b, uint32_t i) { if (b) b[i >> 3] |= (uint8_t)(1u << (i & 7)); }
static int getbit(const uint8_t* b, uint32_t i) { return (b[i >> 3] >> (i & 7)) & 1; }
static void put(Parser* P, uint32_t h, uint32_t pos) { P->table[h] = pos; setbit(P->wr, pos); }
static void emit(Parser* P, uint32_t start, uint32_t lit, uint32_t off, uint32_t len)
{
    if (P->nseq == P->cap) { P->cap = P->cap ? 2 * P->cap : 1024; P->seqs = (Seq*)realloc(P->seqs, P->cap * sizeof(Seq)); }
    P->seqs[P->nseq].start = start; P->seqs[P->nseq].lit = lit; P->seqs[P->nseq].off = off; P->seqs[P->nseq].len = len; P->nseq++;
}

/* runs until the top of the search loop is reached with ip >= stop (returns 1) or the input ends (returns 0) */
static int run(Parser* P, uint32_t stop)
{
    const uint8_t* src = P->src;
    const uint32_t iend = P->n, mfl1 = iend - MFLIMIT + 1, matchlimit = iend - LASTLITERALS;
    for (;;) {
        uint32_t match, ip = P->ip;
        setbit(P->ss, ip);
        if (ip >= stop) return 1;
        {
            uint32_t fwd = ip, step = 1, nb = 1u << SKIPTRIGGER;
            uint32_t fwdH = hash5(src + fwd);
            for (;;) {
                const uint32_t h = fwdH, cur = fwd;
                const uint32_t cand = P->table[h];
                ip = fwd; fwd += step; step = nb++ >> SKIPTRIGGER;
                if (fwd > mfl1) return 0;
                fwdH = hash5(src + fwd);
                put(P, h, cur);
                if (cand < P->floor) continue;                               /* nothing there for a cold parser */
                if (rd32(src + cand) == rd32(src + ip)) {
                    if (cand + DIST_MAX < cur) continue;
                    match = cand; break;
                }
            }
        }
        while (ip > P->anchor && match > P->floor && src[ip - 1] == src[match - 1]) { ip--; match--; }
        {
            uint32_t start = P->anchor, lit = ip - P->anchor;
            for (;;) {                                                       /* next_match */
                uint32_t mc = 0;
                while (ip + MINMATCH + mc < matchlimit && src[ip + MINMATCH + mc] == src[match + MINMATCH + mc]) mc++;
                emit(P, start, lit, ip - match, mc + MINMATCH);
                ip += mc + MINMATCH;
                P->anchor = ip;
                if (ip >= mfl1) { P->ip = ip; return 0; }
                put(P, hash5(src + ip - 2), ip - 2);
                {
                    const uint32_t h = hash5(src + ip), cand = P->table[h];
                    put(P, h, ip);
                    if (cand >= P->floor && cand + DIST_MAX >= ip && rd32(src + cand) == rd32(src + ip)) { match = cand; start = ip; lit = 0; continue; }
                }
                break;
            }
            P->ip = ip + 1;
        }
    }
}

static void parser_init(Parser* P, const uint8_t* src, uint32_t n, uint32_t at, uint8_t* ss, uint8_t* wr)
{
    memset(P, 0, sizeof(*P));
    P->src = src; P->n = n; P->ss = ss; P->wr = wr; P->floor = at;
    for (int i = 0; i < (1 << HASHLOG); ++i) P->table[i] = at;              /* the zero-filled table of a block that starts at `at` */
    put(P, hash5(src + at), at);                                             /* first byte (lz4.c:1443) */
    P->ip = at + 1; P->anchor = at;
}

/* relevant table at p from a write bitmap: for each slot the latest written position in [p - 65535, p) (0xFFFFFFFF: none) */
static void table_at(const uint8_t* src, const uint8_t* wr, uint32_t p, uint32_t lo_limit, uint32_t* T)
{
    memset(T, 0xFF, sizeof(uint32_t) << HASHLOG);
    uint32_t lo = p > DIST_MAX ? p - DIST_MAX : 0;
    if (lo < lo_limit) lo = lo_limit;
    for (uint32_t q = p; q-- > lo; ) if (getbit(wr, q)) { const uint32_t h = hash5(src + q); if (T[h] == 0xFFFFFFFFu) T[h] = q; }
}

/* out[8 * k + ..] per boundary k = 1 .. : B, sync position (0 = none in the zone), strict position (0 = none), actual (1/0),
 * differing relevant slots at the first sync, differing slots at zone end sync, spare, spare.  Returns the number of boundaries. */
int lz4_segment_study(const uint8_t* src, uint32_t n, uint32_t seg, uint32_t warm, uint32_t zone, uint32_t* out, int max_out)
{
    const size_t bm = (size_t)n / 8 + 16;
    uint8_t* ssT = (uint8_t*)calloc(bm, 1); uint8_t* wrT = (uint8_t*)calloc(bm, 1);
    uint8_t* ssC = (uint8_t*)calloc(bm, 1); uint8_t* wrC = (uint8_t*)calloc(bm, 1);
    static uint32_t TT[1 << HASHLOG], TC[1 << HASHLOG];
    Parser T; parser_init(&T, src, n, 0, ssT, wrT);
    run(&T, n);                                                              /* the true parse, once */
    int nb = 0;
    for (uint32_t B = seg; B + zone + 64 < n && nb < max_out; B += seg) {
        const uint32_t at = B > warm ? B - warm : 0;
        Parser C; parser_init(&C, src, n, at, ssC, wrC);
        /* clear this parser's bitmaps over its range */
        memset(ssC + at / 8, 0, (B + zone - at) / 8 + 2); memset(wrC + at / 8, 0, (B + zone - at) / 8 + 2);
        put(&C, hash5(src + at), at);
        run(&C, B + zone);
        uint32_t sync = 0, strict = 0, diff_first = 0, diff_last = 0; int actual = 0;
        for (uint32_t p = B; p < B + zone; ++p) {
            if (!(getbit(ssT, p) && getbit(ssC, p))) continue;
            /* both at the top of the search loop with ip = p; anchors: after a match anchor = ip - 1 in both unless a parser is still in
             * the literal run it started in - then its anchor is further back and the sequences differ: require a match end at p - 1 */
            table_at(src, wrT, p, 0, TT); table_at(src, wrC, p, at, TC);
            uint32_t d = 0;
            for (int i = 0; i < (1 << HASHLOG); ++i) if (TT[i] != TC[i]) d++;
            if (!sync) { sync = p; diff_first = d; }
            diff_last = d;
            if (d == 0) { strict = p; break; }
        }
        /* ground truth: the sequences that start at or after the first sync position agree up to the zone's end */
        if (sync) {
            size_t i = 0, j = 0;
            while (i < T.nseq && T.seqs[i].start + T.seqs[i].lit < sync) i++;
            while (j < C.nseq && C.seqs[j].start + C.seqs[j].lit < sync) j++;
            actual = 1;
            while (i < T.nseq && j < C.nseq && T.seqs[i].start + T.seqs[i].lit < B + zone - 64 && C.seqs[j].start + C.seqs[j].lit < B + zone - 64) {
                const Seq* a = &T.seqs[i]; const Seq* b = &C.seqs[j];
                if (a->start + a->lit != b->start + b->lit || a->off != b->off || a->len != b->len) { actual = 0; break; }
                i++; j++;
            }
        }
        uint32_t* o = out + 8 * nb;
        o[0] = B; o[1] = sync; o[2] = strict; o[3] = (uint32_t)actual; o[4] = diff_first; o[5] = diff_last; o[6] = 0; o[7] = 0;
        nb++;
        free(C.seqs);
    }
    free(T.seqs); free(ssT); free(wrT); free(ssC); free(wrC);
    return nb;
}
