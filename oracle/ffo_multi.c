/*
 * ffo_multi.c -- ORACLE (test infrastructure).  In-process stand-in for an
 * `mpirun -np P fireFoam -parallel` run (reference:
 * cases/wallFireSpread2D/runParallel.sh:12-18, cases/steckler/decompose.sh:2-4):
 * P pthreads, one sub-domain LDU each, the rank-local solver of ffo_solvers.c,
 * block-Jacobi preconditioning, processor-patch exchange and sum-reductions
 * through a pthread barrier.  Reductions add the per-rank
 * partial sums in rank order, so every rank sees the same bits.
 * Also used as the all-core CPU baseline.
 */
#include "ffo.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int P;
    pthread_barrier_t bar;
    double *slots;              /* [P][8] */
    double *const **sendPtrs;   /* [P] -> that rank's send pointer array */
    const int *const *nbrRank, *const *nbrPatch;
} shared_t;

typedef struct {
    shared_t *sh; int rank;
    ffo_ldu *A; int solver, precond; const ffo_controls *k;
    double *psi; const double *source; ffo_perf *perf; int rc;
} task_t;

static void mt_allreduce(void *user, double *vals, int n)
{
    task_t *t = (task_t *)user; shared_t *sh = t->sh;
    if (n > 8) abort();
    for (int i = 0; i < n; i++) sh->slots[t->rank * 8 + i] = vals[i];
    pthread_barrier_wait(&sh->bar);
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int r = 0; r < sh->P; r++) s += sh->slots[r * 8 + i];
        vals[i] = s;
    }
    pthread_barrier_wait(&sh->bar);
}

static void mt_exchange(void *user, int nIf, const int *size, double *const *send, double *const *recv)
{
    task_t *t = (task_t *)user; shared_t *sh = t->sh;
    sh->sendPtrs[t->rank] = send;
    pthread_barrier_wait(&sh->bar);
    for (int p = 0; p < nIf; p++) {
        int nr = sh->nbrRank[t->rank][p], np = sh->nbrPatch[t->rank][p];
        memcpy(recv[p], sh->sendPtrs[nr][np], sizeof(double) * size[p]);
    }
    pthread_barrier_wait(&sh->bar);
}

static void *worker(void *arg)
{
    task_t *t = (task_t *)arg;
    ffo_comm c; c.user = t; c.rank = t->rank; c.nRanks = t->sh->P;
    c.allreduce_sum = mt_allreduce; c.exchange = mt_exchange;
    t->rc = ffo_solve(t->A, t->solver, t->precond, t->k, t->psi, t->source, t->perf, &c);
    return NULL;
}

int ffo_solve_multi(int P, ffo_ldu **A, const int *const *ifNbrRank,
                    const int *const *ifNbrPatch, int solver, int precond,
                    const ffo_controls *k, double **psi,
                    const double *const *source, ffo_perf *perf)
{
    shared_t sh; sh.P = P;
    pthread_barrier_init(&sh.bar, NULL, P);
    sh.slots = (double *)calloc((size_t)P * 8, sizeof(double));
    sh.sendPtrs = (double *const **)calloc(P, sizeof(double *const *));
    sh.nbrRank = ifNbrRank; sh.nbrPatch = ifNbrPatch;
    long g = 0;
    for (int r = 0; r < P; r++) g += A[r]->nCells;
    for (int r = 0; r < P; r++) A[r]->globalCells = g;
    task_t *t = (task_t *)calloc(P, sizeof(task_t));
    pthread_t *th = (pthread_t *)calloc(P, sizeof(pthread_t));
    for (int r = 0; r < P; r++) {
        t[r].sh = &sh; t[r].rank = r; t[r].A = A[r]; t[r].solver = solver; t[r].precond = precond;
        t[r].k = k; t[r].psi = psi[r]; t[r].source = source[r]; t[r].perf = &perf[r];
        pthread_create(&th[r], NULL, worker, &t[r]);
    }
    int rc = 0;
    for (int r = 0; r < P; r++) { pthread_join(th[r], NULL); if (t[r].rc) rc = t[r].rc; }
    pthread_barrier_destroy(&sh.bar);
    free(sh.slots); free((void *)sh.sendPtrs); free(t); free(th);
    return rc;
}
