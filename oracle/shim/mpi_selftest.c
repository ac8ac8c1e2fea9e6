/* TEST INFRASTRUCTURE -- known-answer test of oracle/shim/mpi_shim.c (run by tests/test_grb_shim.py with MPISHIM_NP = 1..4).
 * Checks what main.c relies on: non-overtaking messages per (source, tag) when two tags are in flight, MPI_Waitany
 * over several receives, MPI_Pack / MPI_Unpack round trips, rank-ordered SUM / MAX reductions, time() pinned by
 * MPISHIM_SEED.  Rank 0 prints "mpi selftest ok <np>" when every rank passed. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "mpi.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "rank %d: check failed at line %d: %s\n", me, __LINE__, #c); exit(3); } } while (0)

int main(int argc, char **argv) {
    int np, me;
    MPI_Init(&argc, &argv);
    MPI_Comm_size(MPI_COMM_WORLD, &np);
    MPI_Comm_rank(MPI_COMM_WORLD, &me);
    if (getenv("MPISHIM_SEED")) CHECK(time(NULL) == (time_t) atoll(getenv("MPISHIM_SEED")));

    /* every rank sends every peer TWO messages on tag 1 (first, second) and one on tag 2, packed like main.c:256-259 */
    enum { ROUNDS = 3 };
    char sendbuf[64][3][256], recvbuf[64][3][256];
    MPI_Request sreq[64][3], rreq[64 * 3];
    int nr = 0, src_of[64 * 3], slot_of[64 * 3];
    for (int q = 0; q < np; ++q) {
        if (q == me) continue;
        for (int k = 0; k < ROUNDS; ++k) {
            src_of[nr] = q; slot_of[nr] = k;
            MPI_Irecv(recvbuf[q][k], 256, MPI_PACKED, q, k < 2 ? 1 : 2, MPI_COMM_WORLD, &rreq[nr++]);
        }
    }
    for (int q = 0; q < np; ++q) {
        if (q == me) continue;
        const int order[3] = {2, 0, 1};                       /* tag 2 goes out FIRST, then the two tag-1 messages in order */
        for (int o = 0; o < ROUNDS; ++o) {
            const int k = order[o];
            unsigned long n = 3, idx[3] = {(unsigned long) me, (unsigned long) q, (unsigned long) k};
            float x[3] = {me + 0.5f, q + 0.25f, (float) k};
            int pos = 0;
            MPI_Pack(&n, 1, MPI_UNSIGNED_LONG, sendbuf[q][k], 256, &pos, MPI_COMM_WORLD);
            MPI_Pack(idx, 3, MPI_UNSIGNED_LONG, sendbuf[q][k], 256, &pos, MPI_COMM_WORLD);
            MPI_Pack(x, 3, MPI_FLOAT, sendbuf[q][k], 256, &pos, MPI_COMM_WORLD);
            CHECK(pos == 8 + 24 + 12);                       /* 20 B per (index pair, value) + the count: main.c's wire format */
            MPI_Isend(sendbuf[q][k], pos, MPI_PACKED, q, k < 2 ? 1 : 2, MPI_COMM_WORLD, &sreq[q][k]);
            CHECK(sreq[q][k] != MPI_REQUEST_NULL);
        }
    }
    for (int done = 0; done < nr; ++done) {
        int index;
        MPI_Status st;
        MPI_Waitany(nr, rreq, &index, &st);
        CHECK(index >= 0 && index < nr && rreq[index] == MPI_REQUEST_NULL);
        const int q = src_of[index], k = slot_of[index];
        CHECK(st.MPI_SOURCE == q && st.MPI_TAG == (k < 2 ? 1 : 2));
        unsigned long n, idx[3];
        float x[3];
        int pos = 0;
        MPI_Unpack(recvbuf[q][k], 256, &pos, &n, 1, MPI_UNSIGNED_LONG, MPI_COMM_WORLD);
        MPI_Unpack(recvbuf[q][k], 256, &pos, idx, (int) n, MPI_UNSIGNED_LONG, MPI_COMM_WORLD);
        MPI_Unpack(recvbuf[q][k], 256, &pos, x, (int) n, MPI_FLOAT, MPI_COMM_WORLD);
        /* the receive posted FIRST on tag 1 (slot 0) must hold the message sent first on tag 1 (k = 0) */
        CHECK(n == 3 && idx[0] == (unsigned long) q && idx[1] == (unsigned long) me && idx[2] == (unsigned long) k);
        CHECK(x[0] == q + 0.5f && x[1] == me + 0.25f && x[2] == (float) k);
    }
    {   int index = 7; MPI_Status st; MPI_Waitany(nr, rreq, &index, &st); CHECK(index == -1); }   /* nothing left */
    for (int q = 0; q < np; ++q)
        for (int k = 0; k < ROUNDS && q != me; ++k) { MPI_Status st; MPI_Wait(&sreq[q][k], &st); CHECK(sreq[q][k] == MPI_REQUEST_NULL); }

    /* reductions: ((r0 + r1) + r2) + ... in float -- order matters for these values */
    float v[2] = {me == 0 ? 1e8f : 1.0f, (float) me}, sum[2] = {-1, -1}, want = 1e8f;
    for (int r = 1; r < np; ++r) want = want + 1.0f;          /* 1e8f + 1 == 1e8f in fp32: left-to-right swallows the ones */
    MPI_Allreduce(v, sum, 2, MPI_FLOAT, MPI_SUM, MPI_COMM_WORLD);
    CHECK(sum[0] == want && sum[1] == (float) (np * (np - 1) / 2));
    long long lv = 100 + me, lmax = -1, lsum = -1;
    MPI_Reduce(&lv, &lmax, 1, MPI_LONG_LONG, MPI_MAX, 0, MPI_COMM_WORLD);
    MPI_Reduce(&lv, &lsum, 1, MPI_LONG_LONG, MPI_SUM, 0, MPI_COMM_WORLD);
    if (me == 0) CHECK(lmax == 100 + np - 1 && lsum == 100LL * np + np * (np - 1) / 2);
    else CHECK(lmax == -1 && lsum == -1);                      /* only the root receives */
    double t0 = MPI_Wtime(), tmax = 0;
    MPI_Barrier(MPI_COMM_WORLD);
    double dt = MPI_Wtime() - t0;
    MPI_Reduce(&dt, &tmax, 1, MPI_DOUBLE, MPI_MAX, 0, MPI_COMM_WORLD);
    if (me == 0) { CHECK(tmax >= 0); printf("mpi selftest ok %d\n", np); }
    MPI_Finalize();
    return 0;
}
