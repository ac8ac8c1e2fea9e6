/* TEST INFRASTRUCTURE -- see oracle/shim/mpi.h.  Ranks are forked processes around one shared mapping. */
#define _GNU_SOURCE
#include "mpi.h"
#include <errno.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#define MAX_RANKS 64
#define MAX_MSGS (1 << 16)
#define COLL_SLOT ((size_t) 64 << 20)          /* scratch of one rank in a collective */
#define ARENA_BYTES ((size_t) 16 << 30)        /* message payloads; MAP_NORESERVE, touched pages only */
#define MAX_REQS 4096

struct msg { int src, dst, tag, taken; size_t off, len; };
struct shared {
    pthread_barrier_t bar;
    pthread_mutex_t mtx;
    int nmsg;
    size_t arena_used;
    struct msg msgs[MAX_MSGS];
};
struct req { int active, src, tag; void *buf; size_t cap; unsigned long seq; };

static struct shared *sh = NULL;
static char *arena = NULL, *coll = NULL;
static int np = 1, me = 0;
static pid_t kids[MAX_RANKS];
static struct req reqs[MAX_REQS];
static unsigned long post_seq = 0;           /* order in which this rank posted its receives */
static int req_hi = 0;                       /* one past the highest request slot ever used */
#define SEND_DONE (MAX_REQS + 1)               /* sends complete inside MPI_Isend (the payload is copied) */

static void die(const char *what) { fprintf(stderr, "mpi_shim[%d]: %s\n", me, what); _exit(97); }

static size_t type_size(MPI_Datatype t) {
    switch (t) {
        case MPI_PACKED: return 1;
        case MPI_UNSIGNED_LONG: return sizeof(unsigned long);
        case MPI_FLOAT: return sizeof(float);
        case MPI_DOUBLE: return sizeof(double);
        case MPI_LONG_LONG: return sizeof(long long);
    }
    die("unknown datatype");
    return 0;
}

/* main.c:555 srand(time(NULL)): one seed for all ranks when MPISHIM_SEED is set */
time_t time(time_t *t) {
    const char *s = getenv("MPISHIM_SEED");
    time_t v;
    if (s && *s) v = (time_t) atoll(s);
    else { struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts); v = ts.tv_sec; }
    if (t) *t = v;
    return v;
}

int MPI_Init(int *argc, char ***argv) {
    (void) argc; (void) argv;
    const char *s = getenv("MPISHIM_NP");
    np = s && *s ? atoi(s) : 1;
    if (np < 1 || np > MAX_RANKS) die("MPISHIM_NP out of range");
    sh = (struct shared *) mmap(NULL, sizeof *sh, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    arena = (char *) mmap(NULL, ARENA_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    coll = (char *) mmap(NULL, COLL_SLOT * (size_t) np, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (sh == MAP_FAILED || arena == MAP_FAILED || coll == MAP_FAILED) die("mmap failed");
    pthread_barrierattr_t ba; pthread_barrierattr_init(&ba); pthread_barrierattr_setpshared(&ba, PTHREAD_PROCESS_SHARED);
    pthread_barrier_init(&sh->bar, &ba, (unsigned) np);
    pthread_mutexattr_t ma; pthread_mutexattr_init(&ma); pthread_mutexattr_setpshared(&ma, PTHREAD_PROCESS_SHARED);
    pthread_mutex_init(&sh->mtx, &ma);
    sh->nmsg = 0; sh->arena_used = 0;
    fflush(NULL);
    for (int r = 1; r < np; ++r) {
        pid_t pid = fork();
        if (pid < 0) die("fork failed");
        if (pid == 0) { me = r; break; }
        kids[r] = pid;
    }
    char buf[16];
    snprintf(buf, sizeof buf, "%d", me);
    setenv("MPISHIM_RANK", buf, 1);              /* read by the GraphBLAS shim's dump */
    return MPI_SUCCESS;
}

int MPI_Finalize(void) {
    fflush(NULL);
    pthread_barrier_wait(&sh->bar);
    if (me == 0)
        for (int r = 1; r < np; ++r) {
            int st = 0;
            if (waitpid(kids[r], &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) {
                fprintf(stderr, "mpi_shim: rank %d ended abnormally (status %d)\n", r, st);
                _exit(98);
            }
        }
    return MPI_SUCCESS;
}

int MPI_Comm_size(MPI_Comm comm, int *size) { (void) comm; *size = np; return MPI_SUCCESS; }
int MPI_Comm_rank(MPI_Comm comm, int *rank) { (void) comm; *rank = me; return MPI_SUCCESS; }
double MPI_Wtime(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec; }
int MPI_Barrier(MPI_Comm comm) { (void) comm; pthread_barrier_wait(&sh->bar); return MPI_SUCCESS; }

int MPI_Pack(const void *inbuf, int incount, MPI_Datatype type, void *outbuf, int outsize, int *position, MPI_Comm comm) {
    (void) comm;
    const size_t n = (size_t) incount * type_size(type);
    if ((size_t) *position + n > (size_t) outsize) die("MPI_Pack: buffer too small");
    memcpy((char *) outbuf + *position, inbuf, n);
    *position += (int) n;
    return MPI_SUCCESS;
}

int MPI_Unpack(const void *inbuf, int insize, int *position, void *outbuf, int outcount, MPI_Datatype type, MPI_Comm comm) {
    (void) comm;
    const size_t n = (size_t) outcount * type_size(type);
    if ((size_t) *position + n > (size_t) insize) die("MPI_Unpack: read past the end of the buffer");
    memcpy(outbuf, (const char *) inbuf + *position, n);
    *position += (int) n;
    return MPI_SUCCESS;
}

int MPI_Isend(const void *buf, int count, MPI_Datatype type, int dest, int tag, MPI_Comm comm, MPI_Request *request) {
    (void) comm;
    const size_t n = (size_t) count * type_size(type);
    if (dest < 0 || dest >= np) die("MPI_Isend: bad destination");
    pthread_mutex_lock(&sh->mtx);
    if (sh->nmsg == MAX_MSGS || sh->arena_used + n > ARENA_BYTES) { pthread_mutex_unlock(&sh->mtx); die("MPI_Isend: mailbox full"); }
    const size_t off = sh->arena_used;
    sh->arena_used += (n + 63) & ~(size_t) 63;
    pthread_mutex_unlock(&sh->mtx);
    memcpy(arena + off, buf, n);
    pthread_mutex_lock(&sh->mtx);                /* publish after the payload is in place */
    struct msg *m = &sh->msgs[sh->nmsg];
    m->src = me; m->dst = dest; m->tag = tag; m->taken = 0; m->off = off; m->len = n;
    sh->nmsg++;
    pthread_mutex_unlock(&sh->mtx);
    *request = SEND_DONE;
    return MPI_SUCCESS;
}

int MPI_Irecv(void *buf, int count, MPI_Datatype type, int source, int tag, MPI_Comm comm, MPI_Request *request) {
    (void) comm;
    for (int i = 0; i < MAX_REQS; ++i)
        if (!reqs[i].active) {
            reqs[i].active = 1; reqs[i].src = source; reqs[i].tag = tag; reqs[i].buf = buf;
            reqs[i].cap = (size_t) count * type_size(type);
            reqs[i].seq = ++post_seq;
            if (i + 1 > req_hi) req_hi = i + 1;
            *request = i + 1;
            return MPI_SUCCESS;
        }
    die("MPI_Irecv: out of request slots");
    return 1;
}

/* The earliest message for (source, tag) that nobody took goes to the EARLIEST-POSTED receive for that (source, tag):
 * a receive posted later must not overtake one posted before it (MPI's matching rule; without the check a message that
 * lands between two polls of MPI_Waitany could be taken by the later request). */
static int try_complete(MPI_Request *request, MPI_Status *status) {
    struct req *r = &reqs[*request - 1];
    int hit = 0;
    for (int i = 0; i < req_hi; ++i)
        if (reqs[i].active && reqs[i].src == r->src && reqs[i].tag == r->tag && reqs[i].seq < r->seq) return 0;
    pthread_mutex_lock(&sh->mtx);
    for (int k = 0; k < sh->nmsg; ++k) {
        struct msg *m = &sh->msgs[k];
        if (m->taken || m->dst != me || m->src != r->src || m->tag != r->tag) continue;
        if (m->len > r->cap) { pthread_mutex_unlock(&sh->mtx); die("MPI_Irecv: message longer than the buffer"); }
        m->taken = 1;
        pthread_mutex_unlock(&sh->mtx);
        memcpy(r->buf, arena + m->off, m->len);
        if (status) { status->MPI_SOURCE = m->src; status->MPI_TAG = m->tag; status->MPI_ERROR = MPI_SUCCESS; status->count = (int) m->len; }
        hit = 1;
        break;
    }
    if (!hit) { pthread_mutex_unlock(&sh->mtx); return 0; }
    r->active = 0;
    *request = MPI_REQUEST_NULL;
    return 1;
}

int MPI_Wait(MPI_Request *request, MPI_Status *status) {
    if (*request == MPI_REQUEST_NULL) return MPI_SUCCESS;
    if (*request == SEND_DONE) { *request = MPI_REQUEST_NULL; return MPI_SUCCESS; }
    while (!try_complete(request, status)) sched_yield();
    return MPI_SUCCESS;
}

/* Completes the FIRST pending request of the array (waits for it): a legal MPI_Waitany schedule, and a DETERMINISTIC one.
 * main.c:275-299 adds the product of every received piece in the order MPI_Waitany hands the messages over, so with two
 * or more sources the reference's fp32 sums depend on arrival order (seen here: a 4-rank run did not repeat its own
 * weights bit for bit when completions were taken as they came).  With this schedule the pieces are added by ascending
 * source rank -- the order main.c posts its receives in (:239-243), and the one oracle/pgcn_oracle.c restates. */
int MPI_Waitany(int count, MPI_Request requests[], int *index, MPI_Status *status) {
    for (int i = 0; i < count; ++i) {
        if (requests[i] == MPI_REQUEST_NULL) continue;
        if (requests[i] == SEND_DONE) requests[i] = MPI_REQUEST_NULL;
        else while (!try_complete(&requests[i], status)) sched_yield();
        *index = i;
        return MPI_SUCCESS;
    }
    *index = -1;                                   /* MPI_UNDEFINED: nothing pending */
    return MPI_SUCCESS;
}

static void fold(void *acc, const void *x, int count, MPI_Datatype type, MPI_Op op) {
    for (int i = 0; i < count; ++i) {
#define FOLD(T) { T *a = (T *) acc; const T *b = (const T *) x; a[i] = op == MPI_SUM ? (T) (a[i] + b[i]) : (a[i] > b[i] ? a[i] : b[i]); }
        switch (type) {
            case MPI_FLOAT: FOLD(float) break;
            case MPI_DOUBLE: FOLD(double) break;
            case MPI_LONG_LONG: FOLD(long long) break;
            case MPI_UNSIGNED_LONG: FOLD(unsigned long) break;
            default: die("reduction on an unsupported datatype");
        }
#undef FOLD
    }
}

static int reduce_to(const void *sendbuf, void *recvbuf, int count, MPI_Datatype type, MPI_Op op, int root) {
    const size_t n = (size_t) count * type_size(type);
    if (n > COLL_SLOT) die("collective larger than the scratch slot");
    if (op != MPI_SUM && op != MPI_MAX) die("unsupported reduction");
    memcpy(coll + COLL_SLOT * (size_t) me, sendbuf, n);
    pthread_barrier_wait(&sh->bar);
    if (root < 0 || root == me) {                             /* ((r0 + r1) + r2) + ... on every receiver */
        memcpy(recvbuf, coll, n);
        for (int r = 1; r < np; ++r) fold(recvbuf, coll + COLL_SLOT * (size_t) r, count, type, op);
    }
    pthread_barrier_wait(&sh->bar);
    return MPI_SUCCESS;
}

int MPI_Reduce(const void *sendbuf, void *recvbuf, int count, MPI_Datatype type, MPI_Op op, int root, MPI_Comm comm) {
    (void) comm;
    return reduce_to(sendbuf, recvbuf, count, type, op, root);
}
int MPI_Allreduce(const void *sendbuf, void *recvbuf, int count, MPI_Datatype type, MPI_Op op, MPI_Comm comm) {
    (void) comm;
    return reduce_to(sendbuf, recvbuf, count, type, op, -1);
}
