/* TEST-SIDE stand-in for librccl.so (tests/test_host_cpp.py::test_cpp_tx_rx_bb_two_processes_reduce_and_stop_together): the five entry points libdvbs2hip.so binds
 * with dlopen, for N processes ON ONE GPU -- RCCL itself refuses a communicator with a duplicate device, and this pool has one GPU per box.  The all-reduce goes through a
 * file-backed shared mapping named after the unique id; device buffers are read and written with hipMemcpy on the caller's stream.  Sum of uint64 only (what
 * dvbs2hip_monitor_reduce sends).  Not part of the product: the library finds it only when LD_LIBRARY_PATH points here. */
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

typedef struct { char internal[128]; } nccl_id;
typedef void *hipStream_t;
extern int hipMemcpyAsync(void *dst, const void *src, size_t n, int kind, hipStream_t s);
extern int hipStreamSynchronize(hipStream_t s);

#define MAXR 16
typedef struct {
    volatile uint64_t arrived[2];            /* per parity of the call number */
    volatile uint64_t left[2];
    volatile uint64_t val[2][MAXR][4];
} shm_t;
typedef struct { shm_t *m; int rank, world; uint64_t call; char path[192]; } comm_t;

int ncclGetUniqueId(nccl_id *id)
{
    memset(id, 0, sizeof *id);
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->internal, sizeof id->internal, "/tmp/dvbs2hip_stub_rccl_%d_%lld_%ld", (int)getpid(), (long long)ts.tv_sec, ts.tv_nsec);
    return 0;
}

int ncclCommInitRank(void **comm, int world, nccl_id id, int rank)
{
    if (world < 1 || world > MAXR || rank < 0 || rank >= world) return 4;
    comm_t *c = (comm_t *)calloc(1, sizeof *c);
    c->rank = rank; c->world = world;
    snprintf(c->path, sizeof c->path, "%s", id.internal);
    int fd = open(c->path, O_RDWR | O_CREAT, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(shm_t)) != 0) return 2;
    c->m = (shm_t *)mmap(NULL, sizeof(shm_t), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->m == MAP_FAILED) return 2;
    *comm = c;
    return 0;
}

/* STUB_RCCL_ASYNC=1: like the real library, ncclAllReduce RETURNS once its work is enqueued and the wait for the peers happens ON THE STREAM (a host function that spins in the
 * runtime's callback thread, the result copied to the device behind it) -- so that a peer that dies leaves this rank's stream busy for ever, which is the case
 * dvbs2hip_monitor_reduce's timeout exists for (tests/test_host_cpp.py::test_cpp_tx_rx_bb_leaves_when_its_peer_dies). */
extern int hipLaunchHostFunc(hipStream_t s, void (*fn)(void *), void *user);
extern int hipHostMalloc(void **p, size_t n, unsigned flags);
typedef struct { comm_t *c; int p; size_t count; uint64_t *stage; } wait_t;
static void wait_for_peers(void *u)
{
    wait_t *w = (wait_t *)u;
    shm_t *m = w->c->m;
    for (long spins = 0; m->arrived[w->p] < (uint64_t)w->c->world && spins < 1200000; spins++) usleep(100);      /* at most 120 s */
    uint64_t sum[4] = {0, 0, 0, 0};
    for (int r = 0; r < w->c->world; r++) for (size_t i = 0; i < w->count; i++) sum[i] += m->val[w->p][r][i];
    if (__sync_add_and_fetch(&m->left[w->p], 1) == (uint64_t)w->c->world) { m->left[w->p] = 0; __sync_synchronize(); m->arrived[w->p] = 0; }
    for (size_t i = 0; i < w->count; i++) w->stage[i] = sum[i];
    free(w);
}

int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t s)
{
    comm_t *c = (comm_t *)comm;
    if (dtype != 5 /* ncclUint64 */ || op != 0 /* ncclSum */ || count > 4) return 5;
    uint64_t v[4] = {0, 0, 0, 0};
    if (hipMemcpyAsync(v, send, count * 8, 2 /* D2H */, s) || hipStreamSynchronize(s)) return 1;
    const int p = (int)(c->call++ & 1);
    shm_t *m = c->m;
    if (getenv("STUB_RCCL_ASYNC")) {
        static uint64_t *stage[2] = {NULL, NULL};
        if (!stage[p] && hipHostMalloc((void **)&stage[p], 4 * sizeof(uint64_t), 0)) return 1;
        for (size_t i = 0; i < count; i++) m->val[p][c->rank][i] = v[i];
        __sync_synchronize();
        __sync_fetch_and_add(&m->arrived[p], 1);
        wait_t *w = (wait_t *)malloc(sizeof *w);
        w->c = c; w->p = p; w->count = count; w->stage = stage[p];
        if (hipLaunchHostFunc(s, wait_for_peers, w)) return 1;
        return hipMemcpyAsync(recv, stage[p], count * 8, 1 /* H2D */, s) ? 1 : 0;
    }
    for (size_t i = 0; i < count; i++) m->val[p][c->rank][i] = v[i];
    __sync_synchronize();
    __sync_fetch_and_add(&m->arrived[p], 1);
    for (long spins = 0; m->arrived[p] < (uint64_t)c->world; spins++) { if (spins > 600000) return 6; usleep(100); }      /* 60 s */
    uint64_t sum[4] = {0, 0, 0, 0};
    for (int r = 0; r < c->world; r++) for (size_t i = 0; i < count; i++) sum[i] += m->val[p][r][i];
    /* the last rank to have read resets this parity's counters for the call after the next */
    if (__sync_add_and_fetch(&m->left[p], 1) == (uint64_t)c->world) { m->left[p] = 0; __sync_synchronize(); m->arrived[p] = 0; }
    if (hipMemcpyAsync(recv, sum, count * 8, 1 /* H2D */, s) || hipStreamSynchronize(s)) return 1;
    return 0;
}

int ncclCommDestroy(void *comm)
{
    comm_t *c = (comm_t *)comm;
    if (!c) return 0;
    if (c->rank == 0) unlink(c->path);
    munmap(c->m, sizeof(shm_t));
    free(c);
    return 0;
}

const char *ncclGetErrorString(int e) { return e == 6 ? "stub rccl: a rank did not arrive within 60 s" : e == 5 ? "stub rccl: only sums of <= 4 uint64" : "stub rccl error"; }
