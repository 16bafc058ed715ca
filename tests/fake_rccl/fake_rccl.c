/* fake_rccl.c - TEST INFRASTRUCTURE: a stand-in for librccl.so that moves buffers between PROCESSES THAT SHARE ONE GPU.
 *
 * The sharded product driver (smc.jl_amd/csrc/sharded.hpp, run2.hpp) reaches RCCL through ten entry points it dlopen()s
 * (sharded.hpp:17-48, SMCMI_RCCL_PATH).  Real RCCL needs one GPU per rank, and the GPU boxes of this project have one GPU - so until
 * round 6 the RCCL BRANCH of smcmi_comm_init / smcmi_run_sharded (group start / end, the send / recv counts and displacements of the
 * resample redistribution, the in-stream all-gathers and all-reduces, the mailbox set-up through the communicator) had never executed
 * with more than one rank.  This library exports exactly those ten symbols with RCCL's signatures and semantics as far as the driver
 * uses them, carried over POSIX shared memory: every collective drains the caller's HIP stream, copies device -> shared memory, meets
 * the other ranks at a barrier, copies shared memory -> device.  A call has completed when it returns, which satisfies (trivially)
 * the in-stream ordering real RCCL gives.  It CHECKS what real RCCL would leave to undefined behaviour: every rank must post the same
 * collective with the same count, a recv must find a send of the same count; violations return ncclInvalidUsage instead of hanging.
 * SMCMI_FAKE_RCCL_LOG=<dir>: every rank appends one line per call to <dir>/rank<r>.log (the call-sequence test reads them).
 *
 * Only tests/ loads it (tests/test_gpu_fake_rccl.py); nothing in the product knows it exists.
 * Build: make -C tests/fake_rccl (gcc, libamdhip64 for hipMemcpy / hipStreamSynchronize). */
#define _GNU_SOURCE
#define __HIP_PLATFORM_AMD__ 1
#include <errno.h>
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

enum { OK = 0, UNHANDLED_HIP = 1, SYSTEM_ERROR = 2, INTERNAL = 3, INVALID_ARGUMENT = 4, INVALID_USAGE = 5 };
enum { MAXR = 64, MAX_GROUP_OPS = 4096, NCCL_DOUBLE = 8, NCCL_SUM = 0 };
enum { K_ALLREDUCE = 1, K_ALLGATHER = 2, K_GROUP = 3, K_INIT = 4 };

typedef struct { char internal[128]; } ncclUniqueId;

typedef struct {
    _Atomic uint32_t magic;
    int world;
    _Atomic int arrived;
    _Atomic int generation;
    _Atomic int failed;                           /* a rank met an error: every barrier gives up */
    struct { int kind; long long count; } post[MAXR];
    _Atomic long long cap[MAXR];                  /* bytes of rank r's data segment */
} Ctl;

typedef struct { long long n_msgs; struct { long long dst, count, offset; } m[MAX_GROUP_OPS]; } Dir;   /* head of a data segment during a group */

typedef struct Comm {
    char name[64];
    int rank, world;
    Ctl *ctl;
    char *seg[MAXR];                              /* mappings of the ranks' data segments */
    long long mapped[MAXR];
    FILE *log;
    long long n_calls;
} Comm;

static __thread char g_errbuf[256] = "no error";
static int fail(int code, const char *msg) { snprintf(g_errbuf, sizeof(g_errbuf), "fake RCCL: %s", msg); return code; }

/* ---- group state (per thread, like ncclGroupStart / ncclGroupEnd) */
typedef struct { int is_send, peer; void *buf; long long count; Comm *comm; hipStream_t stream; } GroupOp;
static __thread int g_depth = 0;
static __thread int g_nops = 0;
static __thread GroupOp g_ops[MAX_GROUP_OPS];
static __thread struct Comm *g_last_comm = NULL;   /* the communicator an EMPTY group meets on (a rank without rows to send or receive) */
static __thread hipStream_t g_last_stream = NULL;

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

static int barrier(Comm *c) {
    Ctl *k = c->ctl;
    const int gen = atomic_load(&k->generation);
    if (atomic_fetch_add(&k->arrived, 1) == k->world - 1) {
        atomic_store(&k->arrived, 0);
        atomic_fetch_add(&k->generation, 1);
        return OK;
    }
    const double t0 = now_s();
    double limit = 120.0;
    const char *e = getenv("SMCMI_FAKE_RCCL_TIMEOUT_S");
    if (e && atof(e) > 0.0) limit = atof(e);
    while (atomic_load(&k->generation) == gen) {
        if (atomic_load(&k->failed)) return fail(SYSTEM_ERROR, "another rank failed");
        if (now_s() - t0 > limit) { atomic_store(&k->failed, 1); return fail(SYSTEM_ERROR, "barrier timed out (a rank did not post the collective)"); }
        sched_yield();
    }
    return OK;
}

static void seg_name(const Comm *c, int r, char *out, size_t n) { snprintf(out, n, "%s_d%d", c->name, r); }

/* my data segment holds at least `bytes` (grow only; the new capacity is published before the barrier that lets anybody read) */
static int ensure_own(Comm *c, long long bytes) {
    const int r = c->rank;
    if (bytes <= c->mapped[r]) return OK;
    long long cap = c->mapped[r] > 0 ? c->mapped[r] : (1ll << 16);
    while (cap < bytes) cap *= 2;
    char nm[96];
    seg_name(c, r, nm, sizeof(nm));
    const int fd = shm_open(nm, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return fail(SYSTEM_ERROR, "shm_open (own data segment)");
    if (ftruncate(fd, (off_t)cap) != 0) { close(fd); return fail(SYSTEM_ERROR, "ftruncate (own data segment; /dev/shm full?)"); }
    if (c->seg[r]) munmap(c->seg[r], (size_t)c->mapped[r]);
    c->seg[r] = (char *)mmap(NULL, (size_t)cap, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->seg[r] == MAP_FAILED) { c->seg[r] = NULL; c->mapped[r] = 0; return fail(SYSTEM_ERROR, "mmap (own data segment)"); }
    c->mapped[r] = cap;
    atomic_store(&c->ctl->cap[r], cap);
    return OK;
}
/* rank p's segment as it is now (call after the barrier behind p's post) */
static int map_peer(Comm *c, int p) {
    if (p == c->rank) return OK;
    const long long cap = atomic_load(&c->ctl->cap[p]);
    if (cap <= c->mapped[p]) return OK;
    char nm[96];
    seg_name(c, p, nm, sizeof(nm));
    const int fd = shm_open(nm, O_RDWR, 0600);
    if (fd < 0) return fail(SYSTEM_ERROR, "shm_open (peer data segment)");
    if (c->seg[p]) munmap(c->seg[p], (size_t)c->mapped[p]);
    c->seg[p] = (char *)mmap(NULL, (size_t)cap, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->seg[p] == MAP_FAILED) { c->seg[p] = NULL; c->mapped[p] = 0; return fail(SYSTEM_ERROR, "mmap (peer data segment)"); }
    c->mapped[p] = cap;
    return OK;
}
static int check_posts(Comm *c, int kind, long long count) {
    for (int r = 0; r < c->world; ++r)
        if (c->ctl->post[r].kind != kind || (kind != K_GROUP && c->ctl->post[r].count != count)) {
            atomic_store(&c->ctl->failed, 1);
            char m[160];
            snprintf(m, sizeof(m), "rank %d posted (kind %d, count %lld) while rank %d posted (kind %d, count %lld)", c->rank, kind, count, r,
                     c->ctl->post[r].kind, c->ctl->post[r].count);
            return fail(INVALID_USAGE, m);
        }
    return OK;
}
#define HIPCHK(x) do { if ((x) != hipSuccess) { atomic_store(&c->ctl->failed, 1); return fail(UNHANDLED_HIP, #x); } } while (0)
#define TRY(x) do { int r_ = (x); if (r_ != OK) return r_; } while (0)

const char *ncclGetErrorString(int code) { (void)code; return g_errbuf; }

int ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return fail(INVALID_ARGUMENT, "null id");
    memset(id->internal, 0, sizeof(id->internal));
    struct timespec t;
    clock_gettime(CLOCK_REALTIME, &t);
    snprintf(id->internal, sizeof(id->internal), "/smcmi_fakerccl_%d_%llx", (int)getpid(), (unsigned long long)t.tv_nsec ^ ((unsigned long long)t.tv_sec << 20));
    return OK;
}

int ncclCommInitRank(Comm **out, int nranks, ncclUniqueId id, int rank) {
    if (!out || nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return fail(INVALID_ARGUMENT, "bad (rank, nranks)");
    if (strncmp(id.internal, "/smcmi_fakerccl_", 16) != 0) return fail(INVALID_ARGUMENT, "the unique id does not come from this library");
    Comm *c = (Comm *)calloc(1, sizeof(Comm));
    snprintf(c->name, sizeof(c->name), "%.60s", id.internal);
    c->rank = rank; c->world = nranks;
    int creator = 1;
    int fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 && errno == EEXIST) { creator = 0; fd = shm_open(c->name, O_RDWR, 0600); }
    if (fd < 0) { free(c); return fail(SYSTEM_ERROR, "shm_open (control segment)"); }
    if (creator && ftruncate(fd, (off_t)sizeof(Ctl)) != 0) { close(fd); free(c); return fail(SYSTEM_ERROR, "ftruncate (control segment)"); }
    if (!creator) {                                  /* the creator may not have sized it yet */
        struct stat st;
        const double t0 = now_s();
        for (;;) {
            if (fstat(fd, &st) == 0 && st.st_size >= (off_t)sizeof(Ctl)) break;
            if (now_s() - t0 > 60.0) { close(fd); free(c); return fail(SYSTEM_ERROR, "control segment never sized"); }
            sched_yield();
        }
    }
    c->ctl = (Ctl *)mmap(NULL, sizeof(Ctl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->ctl == MAP_FAILED) { free(c); return fail(SYSTEM_ERROR, "mmap (control segment)"); }
    if (creator) {
        c->ctl->world = nranks;                      /* (a fresh shm segment is zero-filled: counters, flags, capacities start at 0) */
        atomic_store(&c->ctl->magic, 0x5c3171u);
    } else {
        const double t0 = now_s();
        while (atomic_load(&c->ctl->magic) != 0x5c3171u) {
            if (now_s() - t0 > 60.0) { free(c); return fail(SYSTEM_ERROR, "control segment never initialised"); }
            sched_yield();
        }
        if (c->ctl->world != nranks) { free(c); return fail(INVALID_USAGE, "ranks disagree about the communicator's size"); }
    }
    const char *ld = getenv("SMCMI_FAKE_RCCL_LOG");
    if (ld && *ld) {
        char path[512];
        snprintf(path, sizeof(path), "%s/rank%d.log", ld, rank);
        c->log = fopen(path, "a");
        if (c->log) fprintf(c->log, "init world=%d\n", nranks);
    }
    TRY(ensure_own(c, 1 << 16));
    c->ctl->post[rank].kind = K_INIT; c->ctl->post[rank].count = nranks;
    TRY(barrier(c));
    TRY(check_posts(c, K_INIT, nranks));
    TRY(barrier(c));
    if (rank == 0) shm_unlink(c->name);              /* every rank has it mapped: the name can go */
    *out = c;
    g_last_comm = c;
    return OK;
}

int ncclCommDestroy(Comm *c) {
    if (!c) return OK;
    if (g_last_comm == c) g_last_comm = NULL;
    char nm[96];
    seg_name(c, c->rank, nm, sizeof(nm));
    shm_unlink(nm);
    for (int r = 0; r < c->world; ++r)
        if (c->seg[r]) munmap(c->seg[r], (size_t)c->mapped[r]);
    if (c->log) { fprintf(c->log, "destroy calls=%lld\n", c->n_calls); fclose(c->log); }
    munmap(c->ctl, sizeof(Ctl));
    free(c);
    return OK;
}

int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, Comm *c, hipStream_t stream) {
    if (!c || !send || !recv) return fail(INVALID_ARGUMENT, "null argument");
    if (dtype != NCCL_DOUBLE || op != NCCL_SUM) return fail(INVALID_ARGUMENT, "only ncclFloat64 / ncclSum");
    if (g_depth > 0) return fail(INVALID_USAGE, "collectives inside a group are not supported");
    const long long bytes = (long long)count * 8;
    g_last_comm = c; g_last_stream = stream;
    HIPCHK(hipStreamSynchronize(stream));
    TRY(ensure_own(c, bytes));
    HIPCHK(hipMemcpy(c->seg[c->rank], send, (size_t)bytes, hipMemcpyDeviceToHost));
    c->ctl->post[c->rank].kind = K_ALLREDUCE; c->ctl->post[c->rank].count = (long long)count;
    TRY(barrier(c));
    TRY(check_posts(c, K_ALLREDUCE, (long long)count));
    double *tot = (double *)calloc(count ? count : 1, sizeof(double));
    for (int r = 0; r < c->world; ++r) {             /* rank order: the same bits on every rank (as RCCL guarantees for one communicator) */
        int e = map_peer(c, r);
        if (e != OK) { free(tot); return e; }
        const double *x = (const double *)c->seg[r];
        for (size_t k = 0; k < count; ++k) tot[k] += x[k];
    }
    int e = barrier(c);                              /* nobody overwrites its segment before everybody has read it */
    if (e == OK && hipMemcpy(recv, tot, (size_t)bytes, hipMemcpyHostToDevice) != hipSuccess) e = fail(UNHANDLED_HIP, "hipMemcpy (all-reduce result)");
    free(tot);
    if (c->log) fprintf(c->log, "allreduce count=%zu\n", count);
    c->n_calls += 1;
    return e;
}

int ncclAllGather(const void *send, void *recv, size_t sendcount, int dtype, Comm *c, hipStream_t stream) {
    if (!c || !send || !recv) return fail(INVALID_ARGUMENT, "null argument");
    if (dtype != NCCL_DOUBLE) return fail(INVALID_ARGUMENT, "only ncclFloat64");
    if (g_depth > 0) return fail(INVALID_USAGE, "collectives inside a group are not supported");
    const long long bytes = (long long)sendcount * 8;
    g_last_comm = c; g_last_stream = stream;
    HIPCHK(hipStreamSynchronize(stream));
    TRY(ensure_own(c, bytes));
    HIPCHK(hipMemcpy(c->seg[c->rank], send, (size_t)bytes, hipMemcpyDeviceToHost));
    c->ctl->post[c->rank].kind = K_ALLGATHER; c->ctl->post[c->rank].count = (long long)sendcount;
    TRY(barrier(c));
    TRY(check_posts(c, K_ALLGATHER, (long long)sendcount));
    for (int r = 0; r < c->world; ++r) {
        TRY(map_peer(c, r));
        HIPCHK(hipMemcpy((char *)recv + (size_t)r * (size_t)bytes, c->seg[r], (size_t)bytes, hipMemcpyHostToDevice));
    }
    TRY(barrier(c));
    if (c->log) fprintf(c->log, "allgather count=%zu\n", sendcount);
    c->n_calls += 1;
    return OK;
}

int ncclGroupStart(void) { g_depth += 1; return OK; }

static int queue_op(int is_send, void *buf, size_t count, int dtype, int peer, Comm *c, hipStream_t stream) {
    if (!c || (!buf && count)) return fail(INVALID_ARGUMENT, "null argument");
    if (dtype != NCCL_DOUBLE) return fail(INVALID_ARGUMENT, "only ncclFloat64");
    if (peer < 0 || peer >= c->world || peer == c->rank) return fail(INVALID_ARGUMENT, "bad peer (a rank neither sends to nor receives from itself here)");
    if (g_depth == 0) return fail(INVALID_USAGE, "ncclSend / ncclRecv outside ncclGroupStart / ncclGroupEnd: the driver always groups them");
    if (g_nops >= MAX_GROUP_OPS) return fail(INTERNAL, "too many operations in one group");
    GroupOp o = {is_send, peer, buf, (long long)count, c, stream};
    g_ops[g_nops++] = o;
    return OK;
}
int ncclSend(const void *buf, size_t count, int dtype, int peer, Comm *c, hipStream_t stream) { return queue_op(1, (void *)buf, count, dtype, peer, c, stream); }
int ncclRecv(void *buf, size_t count, int dtype, int peer, Comm *c, hipStream_t stream) { return queue_op(0, buf, count, dtype, peer, c, stream); }

int ncclGroupEnd(void) {
    if (g_depth <= 0) return fail(INVALID_USAGE, "ncclGroupEnd without ncclGroupStart");
    if (--g_depth > 0) return OK;
    const int nops = g_nops;
    g_nops = 0;
    /* (a rank whose group is empty - no row of the redistribution starts or ends on it - still meets the others here: the barrier counts
       every rank; real RCCL would simply return) */
    Comm *c = nops ? g_ops[0].comm : g_last_comm;
    if (!c) return OK;
    for (int k = 0; k < nops; ++k)
        if (g_ops[k].comm != c) return fail(INVALID_USAGE, "one communicator per group");
    HIPCHK(hipStreamSynchronize(nops ? g_ops[0].stream : g_last_stream));
    /* my sends -> my segment: [directory | payloads] */
    long long pay = 0, nsend = 0, nrecv = 0;
    for (int k = 0; k < nops; ++k) { if (g_ops[k].is_send) { pay += g_ops[k].count * 8; ++nsend; } else ++nrecv; }
    TRY(ensure_own(c, (long long)sizeof(Dir) + pay));
    Dir *dir = (Dir *)c->seg[c->rank];
    dir->n_msgs = 0;
    long long off = (long long)sizeof(Dir);
    for (int k = 0; k < nops; ++k) {
        if (!g_ops[k].is_send) continue;
        dir->m[dir->n_msgs].dst = g_ops[k].peer; dir->m[dir->n_msgs].count = g_ops[k].count; dir->m[dir->n_msgs].offset = off;
        if (g_ops[k].count) HIPCHK(hipMemcpy(c->seg[c->rank] + off, g_ops[k].buf, (size_t)g_ops[k].count * 8, hipMemcpyDeviceToHost));
        off += g_ops[k].count * 8;
        dir->n_msgs += 1;
    }
    c->ctl->post[c->rank].kind = K_GROUP; c->ctl->post[c->rank].count = nops;
    TRY(barrier(c));
    TRY(check_posts(c, K_GROUP, 0));
    /* my receives: the k-th receive from peer p takes the k-th message p addressed to me */
    int taken[MAXR];
    memset(taken, 0, sizeof(taken));
    int err = OK;
    long long got = 0;
    for (int k = 0; k < nops && err == OK; ++k) {
        if (g_ops[k].is_send) continue;
        const int p = g_ops[k].peer;
        if ((err = map_peer(c, p)) != OK) break;
        const Dir *pd = (const Dir *)c->seg[p];
        long long seen = 0;
        int found = -1;
        for (long long q = 0; q < pd->n_msgs; ++q)
            if (pd->m[q].dst == c->rank) { if (seen == taken[p]) { found = (int)q; break; } ++seen; }
        if (found < 0) { err = fail(INVALID_USAGE, "a receive finds no matching send (the ranks disagree about the redistribution's ranges)"); break; }
        if (pd->m[found].count != g_ops[k].count) { err = fail(INVALID_USAGE, "send and receive counts differ"); break; }
        if (g_ops[k].count && hipMemcpy(g_ops[k].buf, c->seg[p] + pd->m[found].offset, (size_t)g_ops[k].count * 8, hipMemcpyHostToDevice) != hipSuccess)
            err = fail(UNHANDLED_HIP, "hipMemcpy (received rows)");
        taken[p] += 1;
        got += g_ops[k].count;
    }
    /* every message addressed to me must have been received: a send without a receive would hang real RCCL */
    for (int p = 0; p < c->world && err == OK; ++p) {
        if (p == c->rank) continue;
        if ((err = map_peer(c, p)) != OK) break;
        const Dir *pd = (const Dir *)c->seg[p];
        long long to_me = 0;
        for (long long q = 0; q < pd->n_msgs; ++q) to_me += pd->m[q].dst == c->rank;
        if (to_me != taken[p]) err = fail(INVALID_USAGE, "a send finds no matching receive");
    }
    if (err != OK) { atomic_store(&c->ctl->failed, 1); return err; }
    TRY(barrier(c));
    if (c->log) fprintf(c->log, "group sends=%lld recvs=%lld doubles_sent=%lld doubles_received=%lld\n", nsend, nrecv, pay / 8, got);
    c->n_calls += 1;
    return OK;
}
