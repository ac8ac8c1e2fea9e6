/* TEST INFRASTRUCTURE -- not product code.
 *
 * Minimal stand-in for the MPI calls of /root/reference/Parallel-GCN/main.c (no MPI implementation is installed
 * here), so that the reference's unmodified main.c builds as oracle/_ref/grbgcn -- see oracle/shim/GraphBLAS.h.
 * Ranks are PROCESSES: MPI_Init forks MPISHIM_NP - 1 children (default 1 rank) that share one anonymous mapping
 * holding the mailboxes and the scratch of the collectives; rank 0 is the process that was started and waits for
 * the others in MPI_Finalize.  Semantics kept: non-overtaking point-to-point messages per (source, tag),
 * MPI_Waitany over receive requests (completions handed out in request order: deterministic sums, see mpi_shim.c), MPI_Pack / MPI_Unpack as plain byte copies (8 B per MPI_UNSIGNED_LONG,
 * 4 B per MPI_FLOAT: the reference's 20 B per scalar on the wire), reductions in rank order 0, 1, ..., P-1.
 * Also here: time() returns MPISHIM_SEED when that is set, which makes main.c:555 `srand(time(NULL))` -- and with
 * it the weights every rank draws -- reproducible. */
#ifndef PGCN_ORACLE_SHIM_MPI_H
#define PGCN_ORACLE_SHIM_MPI_H

typedef int MPI_Comm;
typedef int MPI_Datatype;
typedef int MPI_Op;
typedef int MPI_Request;
typedef struct { int MPI_SOURCE, MPI_TAG, MPI_ERROR, count; } MPI_Status;

#define MPI_COMM_WORLD 0
#define MPI_SUCCESS 0
#define MPI_REQUEST_NULL 0
enum { MPI_PACKED = 1, MPI_UNSIGNED_LONG, MPI_FLOAT, MPI_DOUBLE, MPI_LONG_LONG };
enum { MPI_SUM = 1, MPI_MAX };

int MPI_Init(int *argc, char ***argv);
int MPI_Finalize(void);
int MPI_Comm_size(MPI_Comm comm, int *size);
int MPI_Comm_rank(MPI_Comm comm, int *rank);
double MPI_Wtime(void);
int MPI_Barrier(MPI_Comm comm);
int MPI_Pack(const void *inbuf, int incount, MPI_Datatype type, void *outbuf, int outsize, int *position, MPI_Comm comm);
int MPI_Unpack(const void *inbuf, int insize, int *position, void *outbuf, int outcount, MPI_Datatype type, MPI_Comm comm);
int MPI_Isend(const void *buf, int count, MPI_Datatype type, int dest, int tag, MPI_Comm comm, MPI_Request *request);
int MPI_Irecv(void *buf, int count, MPI_Datatype type, int source, int tag, MPI_Comm comm, MPI_Request *request);
int MPI_Wait(MPI_Request *request, MPI_Status *status);
int MPI_Waitany(int count, MPI_Request requests[], int *index, MPI_Status *status);
int MPI_Reduce(const void *sendbuf, void *recvbuf, int count, MPI_Datatype type, MPI_Op op, int root, MPI_Comm comm);
int MPI_Allreduce(const void *sendbuf, void *recvbuf, int count, MPI_Datatype type, MPI_Op op, MPI_Comm comm);

#endif
