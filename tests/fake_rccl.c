/* tests/fake_rccl.c -- TEST-ONLY stand-in for librccl.so.1: the entry points csrc/exchange.cpp binds with dlsym, moving
 * bytes between PROCESSES OF ONE HOST through POSIX shared memory (buffers are host memory, streams are ignored), so that the
 * exchange's rank arithmetic, offsets, group structure and error paths run with nranks > 1 on a box without GPUs
 * (tests/test_exchange_fake_rccl.py).  Not part of the product; never installed. */
#define _GNU_SOURCE
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define MAXR 4
#define CAP (1u << 20)
typedef struct { volatile uint64_t full, empty; volatile uint64_t size; unsigned char data[CAP]; } Box;
typedef struct { Box box[MAXR][MAXR]; } Shm;
typedef struct { Shm* shm; int nranks, rank; char name[64]; } Comm;
typedef struct { char internal[128]; } NcclId;
typedef struct { int send; void* buf; size_t bytes; int peer; Comm* c; } Op;

static __thread Op g_ops[64];
static __thread int g_nops = 0, g_group = 0;
static const size_t kTypeSize[16] = {1, 1, 4, 4, 8, 8, 2, 4, 8, 2};

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static int do_send(Comm* c, const void* buf, size_t bytes, int peer)
{
    if (bytes > CAP || peer < 0 || peer >= c->nranks) return 4;      /* ncclInvalidArgument */
    Box* b = &c->shm->box[c->rank][peer];
    const double t0 = now();
    while (b->full != b->empty) { if (now() - t0 > 20.0) return 1; usleep(50); }
    memcpy(b->data, buf, bytes); b->size = bytes;
    __sync_synchronize();
    b->full = b->full + 1;
    return 0;
}
static int do_recv(Comm* c, void* buf, size_t bytes, int peer)
{
    if (bytes > CAP || peer < 0 || peer >= c->nranks) return 4;
    Box* b = &c->shm->box[peer][c->rank];
    const double t0 = now();
    while (b->full == b->empty) { if (now() - t0 > 20.0) return 1; usleep(50); }
    __sync_synchronize();
    if (b->size != bytes) return 5;                                    /* ncclInternalError: the two sides disagree on the size */
    memcpy(buf, b->data, bytes);
    __sync_synchronize();
    b->empty = b->empty + 1;
    return 0;
}
static int run_ops(void)
{   /* a group: every send is posted before any receive is waited for (each ordered pair of ranks has its own box) */
    int rc = 0;
    for (int i = 0; i < g_nops && !rc; ++i) if (g_ops[i].send) rc = do_send(g_ops[i].c, g_ops[i].buf, g_ops[i].bytes, g_ops[i].peer);
    for (int i = 0; i < g_nops && !rc; ++i) if (!g_ops[i].send) rc = do_recv(g_ops[i].c, g_ops[i].buf, g_ops[i].bytes, g_ops[i].peer);
    g_nops = 0;
    return rc;
}
static int post(int send, void* buf, size_t bytes, int peer, Comm* c)
{
    if (!c) return 4;
    if (g_nops == 64) return 5;
    g_ops[g_nops].send = send; g_ops[g_nops].buf = buf; g_ops[g_nops].bytes = bytes; g_ops[g_nops].peer = peer; g_ops[g_nops].c = c; g_nops++;
    return g_group ? 0 : run_ops();
}

int ncclGetUniqueId(NcclId* id)
{
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/fake_rccl_%d_%ld", (int)getpid(), (long)time(NULL));
    return 0;
}
int ncclCommInitRank(Comm** out, int nranks, NcclId id, int rank)
{
    if (nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return 4;
    id.internal[63] = 0;
    int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return 2;
    if (ftruncate(fd, sizeof(Shm)) != 0) { close(fd); return 2; }      /* (new pages read as zero: all boxes empty) */
    Shm* s = (Shm*)mmap(NULL, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (s == MAP_FAILED) return 2;
    Comm* c = (Comm*)calloc(1, sizeof(Comm));
    c->shm = s; c->nranks = nranks; c->rank = rank; strncpy(c->name, id.internal, 63);
    *out = c;
    return 0;
}
int ncclCommDestroy(Comm* c)
{
    if (!c) return 4;
    munmap(c->shm, sizeof(Shm));
    if (c->rank == 0) shm_unlink(c->name);
    free(c);
    return 0;
}
int ncclCommCount(Comm* c, int* n) { if (!c || !n) return 4; *n = c->nranks; return 0; }
int ncclGroupStart(void) { g_group++; return 0; }
int ncclGroupEnd(void) { if (g_group > 0) g_group--; return g_group ? 0 : run_ops(); }
int ncclSend(const void* buf, size_t count, int type, int peer, Comm* c, void* stream) { (void)stream; return post(1, (void*)buf, count * kTypeSize[type & 15], peer, c); }
int ncclRecv(void* buf, size_t count, int type, int peer, Comm* c, void* stream) { (void)stream; return post(0, buf, count * kTypeSize[type & 15], peer, c); }
int ncclAllGather(const void* mine, void* all, size_t count, int type, Comm* c, void* stream)
{
    (void)stream;
    if (!c) return 4;
    const size_t bytes = count * kTypeSize[type & 15];
    int rc = 0;
    memmove((unsigned char*)all + (size_t)c->rank * bytes, mine, bytes);
    for (int d = 1; d < c->nranks && !rc; ++d) rc = do_send(c, mine, bytes, (c->rank + d) % c->nranks);
    for (int d = 1; d < c->nranks && !rc; ++d) { const int from = (c->rank - d + c->nranks) % c->nranks; rc = do_recv(c, (unsigned char*)all + (size_t)from * bytes, bytes, from); }
    return rc;
}
const char* ncclGetErrorString(int rc) { return rc == 1 ? "fake rccl: timeout" : rc == 4 ? "fake rccl: invalid argument" : rc == 5 ? "fake rccl: size mismatch" : "fake rccl: error"; }
