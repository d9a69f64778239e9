/*
 * ffo_ldu.c -- ORACLE (test infrastructure).  LDU matrix container and the
 * lduMatrix kernels, restated from OpenFOAM-dev @940e28f (not in
 * /root/reference):
 *   src/OpenFOAM/matrices/lduMatrix/lduAddressing/lduAddressing.C
 *       (calcLosort, calcOwnerStart, calcLosortStart)
 *   src/OpenFOAM/matrices/lduMatrix/lduMatrix/lduMatrixATmul.C
 *       (Amul, Tmul, sumA, residual)
 *   src/OpenFOAM/matrices/lduMatrix/lduMatrix/lduMatrixSolver.C (normFactor)
 *   src/OpenFOAM/matrices/lduMatrix/lduMatrix/lduMatrixUpdateMatrixInterfaces.C
 *   src/finiteVolume/fields/fvPatchFields/constraint/processor/
 *       processorFvPatchField.C (updateInterfaceMatrix)
 * Reference call sites that reach them: every fvMatrix solve in
 * solver/UEqn.H:19, solver/YEEqn.H:60,111, solver/pEqn.H:39,
 * solver/rhoEqn.H:43; UEqn.H() in solver/pEqn.H:5.
 */
#include "ffo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static void *xmalloc(size_t n) { void *p = malloc(n ? n : 1); if (!p) abort(); return p; }

ffo_ldu *ffo_ldu_create(int nCells, int nFaces, const int *l, const int *u)
{
    ffo_ldu *A = (ffo_ldu *)calloc(1, sizeof(ffo_ldu));
    A->nCells = nCells; A->nFaces = nFaces; A->globalCells = nCells;
    A->l = (int *)xmalloc(sizeof(int) * nFaces);
    A->u = (int *)xmalloc(sizeof(int) * nFaces);
    memcpy(A->l, l, sizeof(int) * nFaces);
    memcpy(A->u, u, sizeof(int) * nFaces);
    for (int f = 0; f < nFaces; f++) {
        if (l[f] < 0 || u[f] >= nCells || l[f] >= u[f]) { ffo_ldu_destroy(A); return NULL; }
        if (f && l[f] < l[f - 1]) { ffo_ldu_destroy(A); return NULL; } /* owner-sorted */
    }
    /* ownerStart: first face owned by each cell (lduAddressing::calcOwnerStart) */
    A->ownerStart = (int *)xmalloc(sizeof(int) * (nCells + 1));
    A->losortStart = (int *)xmalloc(sizeof(int) * (nCells + 1));
    A->losort = (int *)xmalloc(sizeof(int) * nFaces);
    int *cnt = (int *)calloc(nCells + 1, sizeof(int));
    for (int f = 0; f < nFaces; f++) cnt[l[f] + 1]++;
    A->ownerStart[0] = 0;
    for (int c = 0; c < nCells; c++) A->ownerStart[c + 1] = A->ownerStart[c] + cnt[c + 1];
    /* losort: face ids in order of increasing neighbour, stable in face id
     * (lduAddressing::calcLosort)                                           */
    memset(cnt, 0, sizeof(int) * (nCells + 1));
    for (int f = 0; f < nFaces; f++) cnt[u[f] + 1]++;
    A->losortStart[0] = 0;
    for (int c = 0; c < nCells; c++) A->losortStart[c + 1] = A->losortStart[c] + cnt[c + 1];
    int *pos = (int *)xmalloc(sizeof(int) * (nCells + 1));
    memcpy(pos, A->losortStart, sizeof(int) * (nCells + 1));
    for (int f = 0; f < nFaces; f++) A->losort[pos[u[f]]++] = f;
    free(pos); free(cnt);
    A->diag = (double *)calloc(nCells ? nCells : 1, sizeof(double));
    A->upper = (double *)calloc(nFaces ? nFaces : 1, sizeof(double));
    A->lower = A->upper; A->symmetric = 1;
    return A;
}

static void free_interfaces(ffo_ldu *A)
{
    for (int p = 0; p < A->nIf; p++) {
        free(A->ifFaceCells[p]); free(A->ifBouCoeffs[p]); free(A->ifIntCoeffs[p]);
        free(A->ifSend[p]); free(A->ifRecv[p]);
    }
    free(A->ifSize); free(A->ifFaceCells); free(A->ifBouCoeffs); free(A->ifIntCoeffs);
    free(A->ifSend); free(A->ifRecv);
    A->nIf = 0; A->ifSize = NULL; A->ifFaceCells = NULL; A->ifBouCoeffs = NULL;
    A->ifIntCoeffs = NULL; A->ifSend = A->ifRecv = NULL;
}

void ffo_ldu_destroy(ffo_ldu *A)
{
    if (!A) return;
    free_interfaces(A);
    free(A->l); free(A->u); free(A->ownerStart); free(A->losort); free(A->losortStart);
    if (A->lower != A->upper) free(A->lower);
    free(A->upper); free(A->diag); free(A);
}

void ffo_ldu_set_coeffs(ffo_ldu *A, const double *diag, const double *upper,
                        const double *lower)
{
    memcpy(A->diag, diag, sizeof(double) * A->nCells);
    memcpy(A->upper, upper, sizeof(double) * A->nFaces);
    if (lower) {
        if (A->lower == A->upper) A->lower = (double *)xmalloc(sizeof(double) * A->nFaces);
        memcpy(A->lower, lower, sizeof(double) * A->nFaces);
        A->symmetric = 0;
    } else {
        if (A->lower != A->upper) free(A->lower);
        A->lower = A->upper; A->symmetric = 1;
    }
}

void ffo_ldu_set_global_cells(ffo_ldu *A, long g) { A->globalCells = g; }

void ffo_ldu_set_interfaces(ffo_ldu *A, int nIf, const int *size,
                            const int *const *faceCells,
                            const double *const *bouCoeffs,
                            const double *const *intCoeffs)
{
    free_interfaces(A);
    A->nIf = nIf;
    A->ifSize = (int *)xmalloc(sizeof(int) * nIf);
    A->ifFaceCells = (int **)xmalloc(sizeof(int *) * nIf);
    A->ifBouCoeffs = (double **)xmalloc(sizeof(double *) * nIf);
    A->ifIntCoeffs = (double **)xmalloc(sizeof(double *) * nIf);
    A->ifSend = (double **)xmalloc(sizeof(double *) * nIf);
    A->ifRecv = (double **)xmalloc(sizeof(double *) * nIf);
    for (int p = 0; p < nIf; p++) {
        int n = size[p];
        A->ifSize[p] = n;
        A->ifFaceCells[p] = (int *)xmalloc(sizeof(int) * n);
        A->ifBouCoeffs[p] = (double *)xmalloc(sizeof(double) * n);
        A->ifIntCoeffs[p] = (double *)xmalloc(sizeof(double) * n);
        A->ifSend[p] = (double *)xmalloc(sizeof(double) * n);
        A->ifRecv[p] = (double *)xmalloc(sizeof(double) * n);
        memcpy(A->ifFaceCells[p], faceCells[p], sizeof(int) * n);
        memcpy(A->ifBouCoeffs[p], bouCoeffs[p], sizeof(double) * n);
        if (intCoeffs && intCoeffs[p]) memcpy(A->ifIntCoeffs[p], intCoeffs[p], sizeof(double) * n);
        else memcpy(A->ifIntCoeffs[p], bouCoeffs[p], sizeof(double) * n);
    }
}

/* initMatrixInterfaces + updateMatrixInterfaces for processor patches:
 *   pnf = neighbour rank's psi[faceCells]           (patchNeighbourField)
 *   result[faceCells[i]] -= coeffs[i]*pnf[i]        (updateInterfaceMatrix) */
static void update_interfaces(const ffo_ldu *A, double *const *coeffs,
                              const double *x, double *y, const ffo_comm *c)
{
    if (!A->nIf) return;
    for (int p = 0; p < A->nIf; p++)
        for (int i = 0; i < A->ifSize[p]; i++) A->ifSend[p][i] = x[A->ifFaceCells[p][i]];
    if (c && c->exchange) c->exchange(c->user, A->nIf, A->ifSize, A->ifSend, A->ifRecv);
    else abort(); /* coupled interfaces need a communicator */
    for (int p = 0; p < A->nIf; p++)
        for (int i = 0; i < A->ifSize[p]; i++)
            y[A->ifFaceCells[p][i]] -= coeffs[p][i] * A->ifRecv[p][i];
}

/* lduMatrix::Amul */
void ffo_amul(const ffo_ldu *A, const double *x, double *y, const ffo_comm *c)
{
    const int *l = A->l, *u = A->u;
    for (int i = 0; i < A->nCells; i++) y[i] = A->diag[i] * x[i];
    for (int f = 0; f < A->nFaces; f++) {
        y[u[f]] += A->lower[f] * x[l[f]];
        y[l[f]] += A->upper[f] * x[u[f]];
    }
    update_interfaces(A, A->ifBouCoeffs, x, y, c);
}

/* lduMatrix::Tmul (interfaces use interfaceIntCoeffs) */
void ffo_tmul(const ffo_ldu *A, const double *x, double *y, const ffo_comm *c)
{
    const int *l = A->l, *u = A->u;
    for (int i = 0; i < A->nCells; i++) y[i] = A->diag[i] * x[i];
    for (int f = 0; f < A->nFaces; f++) {
        y[u[f]] += A->upper[f] * x[l[f]];
        y[l[f]] += A->lower[f] * x[u[f]];
    }
    update_interfaces(A, A->ifIntCoeffs, x, y, c);
}

/* lduMatrix::sumA */
void ffo_sumA(const ffo_ldu *A, double *s)
{
    const int *l = A->l, *u = A->u;
    for (int i = 0; i < A->nCells; i++) s[i] = A->diag[i];
    for (int f = 0; f < A->nFaces; f++) {
        s[u[f]] += A->lower[f];
        s[l[f]] += A->upper[f];
    }
    for (int p = 0; p < A->nIf; p++)
        for (int i = 0; i < A->ifSize[p]; i++) s[A->ifFaceCells[p][i]] -= A->ifBouCoeffs[p][i];
}

/* lduMatrix::residual:  r = b - A x, built with the loop order of the
 * upstream routine (rA = source - diag*psi; then face loop subtracting).    */
void ffo_residual(const ffo_ldu *A, const double *x, const double *b,
                  double *r, const ffo_comm *c)
{
    const int *l = A->l, *u = A->u;
    /* upstream negates the interface coefficients and adds; equivalent to
     * subtracting A's interface term: r[fc] += bou*pnf                      */
    for (int i = 0; i < A->nCells; i++) r[i] = b[i] - A->diag[i] * x[i];
    for (int f = 0; f < A->nFaces; f++) {
        r[u[f]] -= A->lower[f] * x[l[f]];
        r[l[f]] -= A->upper[f] * x[u[f]];
    }
    if (A->nIf) {
        for (int p = 0; p < A->nIf; p++)
            for (int i = 0; i < A->ifSize[p]; i++) A->ifSend[p][i] = x[A->ifFaceCells[p][i]];
        if (c && c->exchange) c->exchange(c->user, A->nIf, A->ifSize, A->ifSend, A->ifRecv);
        else abort();
        for (int p = 0; p < A->nIf; p++)
            for (int i = 0; i < A->ifSize[p]; i++)
                r[A->ifFaceCells[p][i]] += A->ifBouCoeffs[p][i] * A->ifRecv[p][i];
    }
}

/* lduMatrix::solver::normFactor:
 *   tmp = sumA * gAverage(psi);
 *   return gSum(|Apsi - tmp| + |source - tmp|) + solverPerformance::small_  */
double ffo_norm_factor(const ffo_ldu *A, const double *x, const double *b,
                       const double *Ax, double *tmp, const ffo_comm *c)
{
    double red[2];
    double sx = 0.0;
    for (int i = 0; i < A->nCells; i++) sx += x[i];
    red[0] = sx;
    if (c && c->allreduce_sum) c->allreduce_sum(c->user, red, 1);
    double xRef = red[0] / (double)A->globalCells;
    ffo_sumA(A, tmp);
    for (int i = 0; i < A->nCells; i++) tmp[i] *= xRef;
    double s = 0.0;
    for (int i = 0; i < A->nCells; i++) s += fabs(Ax[i] - tmp[i]) + fabs(b[i] - tmp[i]);
    red[0] = s;
    if (c && c->allreduce_sum) c->allreduce_sum(c->user, red, 1);
    return red[0] + 1e-20;
}

/* SURVEY 8(d) counter-based hash: splitmix64 finaliser of (seed ^ idx)      */
double ffo_hash_u(uint64_t seed, uint64_t idx)
{
    uint64_t z = (seed ^ idx) + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

/* blockMesh single block, SURVEY A.1: cell c = i + nx*(j + ny*k); internal
 * faces in upper-triangular order: for c ascending emit (c,c+1), (c,c+nx),
 * (c,c+nx*ny) when they exist.                                              */
void ffo_hex_counts(int nx, int ny, int nz, long *nCells, long *nFaces)
{
    *nCells = (long)nx * ny * nz;
    *nFaces = (long)(nx - 1) * ny * nz + (long)nx * (ny - 1) * nz + (long)nx * ny * (nz - 1);
}

void ffo_hex_ldu(int nx, int ny, int nz, int *l, int *u)
{
    long f = 0;
    for (int k = 0; k < nz; k++)
        for (int j = 0; j < ny; j++)
            for (int i = 0; i < nx; i++) {
                int c = i + nx * (j + ny * k);
                if (i < nx - 1) { l[f] = c; u[f] = c + 1; f++; }
                if (j < ny - 1) { l[f] = c; u[f] = c + nx; f++; }
                if (k < nz - 1) { l[f] = c; u[f] = c + nx * ny; f++; }
            }
}
