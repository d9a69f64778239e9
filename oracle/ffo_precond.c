/*
 * ffo_precond.c -- ORACLE (test infrastructure).  Serial preconditioner and
 * smoother sweeps in the exact face order of OpenFOAM-dev @940e28f (not in
 * /root/reference):
 *   src/OpenFOAM/matrices/lduMatrix/preconditioners/DICPreconditioner/DICPreconditioner.C
 *   src/OpenFOAM/matrices/lduMatrix/preconditioners/DILUPreconditioner/DILUPreconditioner.C
 *   src/OpenFOAM/matrices/lduMatrix/smoothers/GaussSeidel/GaussSeidelSmoother.C
 *   src/OpenFOAM/matrices/lduMatrix/smoothers/symGaussSeidel/symGaussSeidelSmoother.C
 * Selected by the reference in cases/steckler/system/fvSolution:21-61
 * (PCG+DIC for p_rgh/ph_rgh/rho, smoothSolver+symGaussSeidel for U/Yi/h/k) and
 * cases/wallFireSpread2D/system/fvSolution:67-75,115-152 (PBiCG+DILU).
 * Across ranks these are block-Jacobi: processor-interface coefficients are
 * ignored by DIC/DILU; Gauss-Seidel treats them explicitly (lagged).
 */
#include "ffo.h"
#include <stdlib.h>
#include <string.h>

/* DICPreconditioner::calcReciprocalD */
void ffo_dic_calc_rD(const ffo_ldu *A, double *rD)
{
    const int *l = A->l, *u = A->u; const double *up = A->upper;
    for (int i = 0; i < A->nCells; i++) rD[i] = A->diag[i];
    for (int f = 0; f < A->nFaces; f++) rD[u[f]] -= up[f] * up[f] / rD[l[f]];
    for (int i = 0; i < A->nCells; i++) rD[i] = 1.0 / rD[i];
}

/* DICPreconditioner::precondition */
void ffo_dic_precondition(const ffo_ldu *A, const double *rD, const double *r, double *w)
{
    const int *l = A->l, *u = A->u; const double *up = A->upper;
    for (int i = 0; i < A->nCells; i++) w[i] = rD[i] * r[i];
    for (int f = 0; f < A->nFaces; f++) w[u[f]] -= rD[u[f]] * up[f] * w[l[f]];
    for (int f = A->nFaces - 1; f >= 0; f--) w[l[f]] -= rD[l[f]] * up[f] * w[u[f]];
}

/* DILUPreconditioner::calcReciprocalD */
void ffo_dilu_calc_rD(const ffo_ldu *A, double *rD)
{
    const int *l = A->l, *u = A->u;
    for (int i = 0; i < A->nCells; i++) rD[i] = A->diag[i];
    for (int f = 0; f < A->nFaces; f++) rD[u[f]] -= A->upper[f] * A->lower[f] / rD[l[f]];
    for (int i = 0; i < A->nCells; i++) rD[i] = 1.0 / rD[i];
}

/* DILUPreconditioner::precondition */
void ffo_dilu_precondition(const ffo_ldu *A, const double *rD, const double *r, double *w)
{
    const int *l = A->l, *u = A->u, *lo = A->losort;
    for (int i = 0; i < A->nCells; i++) w[i] = rD[i] * r[i];
    for (int k = 0; k < A->nFaces; k++) {
        int sf = lo[k];
        w[u[sf]] -= rD[u[sf]] * A->lower[sf] * w[l[sf]];
    }
    for (int f = A->nFaces - 1; f >= 0; f--) w[l[f]] -= rD[l[f]] * A->upper[f] * w[u[f]];
}

/* DILUPreconditioner::preconditionT */
void ffo_dilu_preconditionT(const ffo_ldu *A, const double *rD, const double *r, double *w)
{
    const int *l = A->l, *u = A->u, *lo = A->losort;
    for (int i = 0; i < A->nCells; i++) w[i] = rD[i] * r[i];
    for (int f = 0; f < A->nFaces; f++) w[u[f]] -= rD[u[f]] * A->upper[f] * w[l[f]];
    for (int k = A->nFaces - 1; k >= 0; k--) {
        int sf = lo[k];
        w[l[sf]] -= rD[l[sf]] * A->lower[sf] * w[u[sf]];
    }
}

/* GaussSeidelSmoother::smooth (symmetric_sweep = 0) and
 * symGaussSeidelSmoother::smooth (symmetric_sweep = 1).
 * Per sweep: bPrime = source; coupled boundaries are added explicitly with
 * the negated interfaceBouCoeffs (bPrime[fc] += bou*pnf, psi lagged);
 * forward row sweep; for symGS a reverse row sweep with the same bPrime.    */
void ffo_gs_smooth(const ffo_ldu *A, double *psi, const double *b, int nSweeps,
                   int symmetric_sweep, const ffo_comm *c)
{
    const int n = A->nCells; const int *u = A->u, *os = A->ownerStart;
    const double *up = A->upper, *lw = A->lower, *dg = A->diag;
    double *bP = (double *)malloc(sizeof(double) * (n ? n : 1));
    for (int s = 0; s < nSweeps; s++) {
        memcpy(bP, b, sizeof(double) * n);
        if (A->nIf) {
            for (int p = 0; p < A->nIf; p++)
                for (int i = 0; i < A->ifSize[p]; i++) A->ifSend[p][i] = psi[A->ifFaceCells[p][i]];
            if (c && c->exchange) c->exchange(c->user, A->nIf, A->ifSize, A->ifSend, A->ifRecv);
            else abort();
            for (int p = 0; p < A->nIf; p++)
                for (int i = 0; i < A->ifSize[p]; i++)
                    bP[A->ifFaceCells[p][i]] += A->ifBouCoeffs[p][i] * A->ifRecv[p][i];
        }
        for (int ci = 0; ci < n; ci++) {
            double psii = bP[ci];
            for (int f = os[ci]; f < os[ci + 1]; f++) psii -= up[f] * psi[u[f]];
            psii /= dg[ci];
            for (int f = os[ci]; f < os[ci + 1]; f++) bP[u[f]] -= lw[f] * psii;
            psi[ci] = psii;
        }
        if (symmetric_sweep) {
            for (int ci = n - 1; ci >= 0; ci--) {
                double psii = bP[ci];
                for (int f = os[ci]; f < os[ci + 1]; f++) psii -= up[f] * psi[u[f]];
                psii /= dg[ci];
                for (int f = os[ci]; f < os[ci + 1]; f++) bP[u[f]] -= lw[f] * psii;
                psi[ci] = psii;
            }
        }
    }
    free(bP);
}
