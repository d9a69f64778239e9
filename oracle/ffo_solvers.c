/*
 * ffo_solvers.c -- ORACLE (test infrastructure).  Krylov / smooth / diagonal
 * solver drivers restated from OpenFOAM-dev @940e28f (not in /root/reference;
 * SURVEY Appendix A.5):
 *   src/OpenFOAM/matrices/lduMatrix/solvers/PCG/PCG.C
 *   src/OpenFOAM/matrices/lduMatrix/solvers/PBiCGStab/PBiCGStab.C
 *   src/OpenFOAM/matrices/lduMatrix/solvers/PBiCG/PBiCG.C
 *   src/OpenFOAM/matrices/lduMatrix/solvers/smoothSolver/smoothSolver.C
 *   src/OpenFOAM/matrices/lduMatrix/solvers/diagonalSolver/diagonalSolver.C
 *   src/OpenFOAM/matrices/LduMatrix/LduMatrix/SolverPerformance.C
 *       (checkConvergence, checkSingularity; small_=1e-20, vsmall_=1e-300)
 * Reference selection: cases/steckler/system/fvSolution:21-61 (PCG/DIC,
 * smoothSolver/symGaussSeidel maxIter 10), cases/wallFireSpread2D/system/
 * fvSolution:115-152 (PBiCG/DILU), north_star (PBiCGStab/DILU); call sites
 * solver/pEqn.H:39, solver/UEqn.H:19, solver/YEEqn.H:60,111,
 * solver/rhoEqn.H:43, solver/phrghEqn.H:48.
 * Defaults (lduMatrix::solver::readControls): maxIter 1000, minIter 0,
 * tolerance 1e-6, relTol 0.
 */
#include "ffo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SP_SMALL 1e-20
#define SP_VSMALL 1e-300
#define SP_GREAT 1e+20

static double gsum(double v, const ffo_comm *c)
{
    if (c && c->allreduce_sum) c->allreduce_sum(c->user, &v, 1);
    return v;
}
static double gSumProd(int n, const double *a, const double *b, const ffo_comm *c)
{ double s = 0.0; for (int i = 0; i < n; i++) s += a[i] * b[i]; return gsum(s, c); }
static double gSumMag(int n, const double *a, const ffo_comm *c)
{ double s = 0.0; for (int i = 0; i < n; i++) s += fabs(a[i]); return gsum(s, c); }
static double gSumSqr(int n, const double *a, const ffo_comm *c)
{ double s = 0.0; for (int i = 0; i < n; i++) s += a[i] * a[i]; return gsum(s, c); }

static int check_convergence(ffo_perf *p, double tol, double relTol)
{
    if (p->finalResidual < tol ||
        (relTol > SP_SMALL && p->finalResidual < relTol * p->initialResidual))
        p->converged = 1;
    else
        p->converged = 0;
    return p->converged;
}
static int check_singularity(ffo_perf *p, double residual)
{
    p->singular = (residual > SP_VSMALL) ? 0 : 1;
    return p->singular;
}

typedef void (*precond_fn)(const ffo_ldu *, const double *, const double *, double *);
static void no_precond(const ffo_ldu *A, const double *rD, const double *r, double *w)
{ (void)rD; memcpy(w, r, sizeof(double) * A->nCells); }
static void diag_precond(const ffo_ldu *A, const double *rD, const double *r, double *w)
{ for (int i = 0; i < A->nCells; i++) w[i] = rD[i] * r[i]; }

static int setup_precond(const ffo_ldu *A, int precond, double *rD, precond_fn *fn, precond_fn *fnT)
{
    switch (precond) {
    case FFO_NONE: *fn = no_precond; *fnT = no_precond; return 0;
    case FFO_DIC:  ffo_dic_calc_rD(A, rD); *fn = ffo_dic_precondition; *fnT = ffo_dic_precondition; return 0;
    case FFO_DILU: ffo_dilu_calc_rD(A, rD); *fn = ffo_dilu_precondition; *fnT = ffo_dilu_preconditionT; return 0;
    case FFO_DIAGONALP:
        for (int i = 0; i < A->nCells; i++) rD[i] = 1.0 / A->diag[i];
        *fn = diag_precond; *fnT = diag_precond; return 0;
    }
    return -1;
}

/* PCG::solve */
static int pcg(const ffo_ldu *A, int precond, const ffo_controls *k, double *psi,
               const double *source, ffo_perf *perf, const ffo_comm *c)
{
    const int n = A->nCells;
    double *pA = (double *)malloc(sizeof(double) * (n + 1)), *wA = (double *)malloc(sizeof(double) * (n + 1));
    double *rA = (double *)malloc(sizeof(double) * (n + 1)), *rD = (double *)malloc(sizeof(double) * (n + 1));
    double wArA = SP_GREAT, wArAold = wArA;
    ffo_amul(A, psi, wA, c);
    for (int i = 0; i < n; i++) rA[i] = source[i] - wA[i];
    double normFactor = ffo_norm_factor(A, psi, source, wA, pA, c);
    perf->initialResidual = gSumMag(n, rA, c) / normFactor;
    perf->finalResidual = perf->initialResidual;
    perf->nIterations = 0; perf->singular = 0;
    int rc = 0;
    if (k->minIter > 0 || !check_convergence(perf, k->tolerance, k->relTol)) {
        precond_fn P, PT;
        rc = setup_precond(A, precond, rD, &P, &PT);
        if (!rc) do {
            wArAold = wArA;
            P(A, rD, rA, wA);
            wArA = gSumProd(n, wA, rA, c);
            if (perf->nIterations == 0) {
                for (int i = 0; i < n; i++) pA[i] = wA[i];
            } else {
                double beta = wArA / wArAold;
                for (int i = 0; i < n; i++) pA[i] = wA[i] + beta * pA[i];
            }
            ffo_amul(A, pA, wA, c);
            double wApA = gSumProd(n, wA, pA, c);
            if (check_singularity(perf, fabs(wApA) / normFactor)) break;
            double alpha = wArA / wApA;
            for (int i = 0; i < n; i++) { psi[i] += alpha * pA[i]; rA[i] -= alpha * wA[i]; }
            perf->finalResidual = gSumMag(n, rA, c) / normFactor;
        } while ((++perf->nIterations < k->maxIter &&
                  !check_convergence(perf, k->tolerance, k->relTol)) ||
                 perf->nIterations < k->minIter);
    }
    free(pA); free(wA); free(rA); free(rD);
    return rc;
}

/* PBiCGStab::solve */
static int pbicgstab(const ffo_ldu *A, int precond, const ffo_controls *k, double *psi,
                     const double *source, ffo_perf *perf, const ffo_comm *c)
{
    const int n = A->nCells; const size_t sz = sizeof(double) * (n + 1);
    double *yA = (double *)malloc(sz), *rA = (double *)malloc(sz), *pA = (double *)malloc(sz);
    double *rD = (double *)malloc(sz);
    ffo_amul(A, psi, yA, c);
    for (int i = 0; i < n; i++) rA[i] = source[i] - yA[i];
    double normFactor = ffo_norm_factor(A, psi, source, yA, pA, c);
    perf->initialResidual = gSumMag(n, rA, c) / normFactor;
    perf->finalResidual = perf->initialResidual;
    perf->nIterations = 0; perf->singular = 0;
    int rc = 0;
    if (k->minIter > 0 || !check_convergence(perf, k->tolerance, k->relTol)) {
        double *AyA = (double *)malloc(sz), *sA = (double *)malloc(sz), *zA = (double *)malloc(sz);
        double *tA = (double *)malloc(sz), *rA0 = (double *)malloc(sz);
        memcpy(rA0, rA, sizeof(double) * n);
        double rA0rA = 0, alpha = 0, omega = 0;
        precond_fn P, PT;
        rc = setup_precond(A, precond, rD, &P, &PT);
        if (!rc) do {
            const double rA0rAold = rA0rA;
            rA0rA = gSumProd(n, rA0, rA, c);
            if (check_singularity(perf, fabs(rA0rA))) break;
            if (perf->nIterations == 0) {
                for (int i = 0; i < n; i++) pA[i] = rA[i];
            } else {
                if (check_singularity(perf, fabs(omega))) break;
                const double beta = (rA0rA / rA0rAold) * (alpha / omega);
                for (int i = 0; i < n; i++) pA[i] = rA[i] + beta * (pA[i] - omega * AyA[i]);
            }
            P(A, rD, pA, yA);
            ffo_amul(A, yA, AyA, c);
            const double rA0AyA = gSumProd(n, rA0, AyA, c);
            alpha = rA0rA / rA0AyA;
            for (int i = 0; i < n; i++) sA[i] = rA[i] - alpha * AyA[i];
            perf->finalResidual = gSumMag(n, sA, c) / normFactor;
            if (check_convergence(perf, k->tolerance, k->relTol)) {
                for (int i = 0; i < n; i++) psi[i] += alpha * yA[i];
                perf->nIterations++;
                free(AyA); free(sA); free(zA); free(tA); free(rA0);
                free(yA); free(rA); free(pA); free(rD);
                return 0;
            }
            P(A, rD, sA, zA);
            ffo_amul(A, zA, tA, c);
            const double tAtA = gSumSqr(n, tA, c);
            omega = gSumProd(n, tA, sA, c) / tAtA;
            for (int i = 0; i < n; i++) {
                psi[i] += alpha * yA[i] + omega * zA[i];
                rA[i] = sA[i] - omega * tA[i];
            }
            perf->finalResidual = gSumMag(n, rA, c) / normFactor;
        } while ((++perf->nIterations < k->maxIter &&
                  !check_convergence(perf, k->tolerance, k->relTol)) ||
                 perf->nIterations < k->minIter);
        free(AyA); free(sA); free(zA); free(tA); free(rA0);
    }
    free(yA); free(rA); free(pA); free(rD);
    return rc;
}

/* PBiCG::solve */
static int pbicg(const ffo_ldu *A, int precond, const ffo_controls *k, double *psi,
                 const double *source, ffo_perf *perf, const ffo_comm *c)
{
    const int n = A->nCells; const size_t sz = sizeof(double) * (n + 1);
    double *pA = (double *)malloc(sz), *pT = (double *)calloc(n + 1, sizeof(double));
    double *wA = (double *)malloc(sz), *wT = (double *)malloc(sz);
    double *rA = (double *)malloc(sz), *rT = (double *)malloc(sz), *rD = (double *)malloc(sz);
    double wArT = SP_GREAT, wArTold = wArT;
    ffo_amul(A, psi, wA, c);
    ffo_tmul(A, psi, wT, c);
    for (int i = 0; i < n; i++) { rA[i] = source[i] - wA[i]; rT[i] = source[i] - wT[i]; }
    double normFactor = ffo_norm_factor(A, psi, source, wA, pA, c);
    perf->initialResidual = gSumMag(n, rA, c) / normFactor;
    perf->finalResidual = perf->initialResidual;
    perf->nIterations = 0; perf->singular = 0;
    int rc = 0;
    if (k->minIter > 0 || !check_convergence(perf, k->tolerance, k->relTol)) {
        precond_fn P, PT;
        rc = setup_precond(A, precond, rD, &P, &PT);
        if (!rc) do {
            wArTold = wArT;
            P(A, rD, rA, wA);
            PT(A, rD, rT, wT);
            wArT = gSumProd(n, wA, rT, c);
            if (perf->nIterations == 0) {
                for (int i = 0; i < n; i++) { pA[i] = wA[i]; pT[i] = wT[i]; }
            } else {
                double beta = wArT / wArTold;
                for (int i = 0; i < n; i++) { pA[i] = wA[i] + beta * pA[i]; pT[i] = wT[i] + beta * pT[i]; }
            }
            ffo_amul(A, pA, wA, c);
            ffo_tmul(A, pT, wT, c);
            double wApT = gSumProd(n, wA, pT, c);
            if (check_singularity(perf, fabs(wApT) / normFactor)) break;
            double alpha = wArT / wApT;
            for (int i = 0; i < n; i++) {
                psi[i] += alpha * pA[i]; rA[i] -= alpha * wA[i]; rT[i] -= alpha * wT[i];
            }
            perf->finalResidual = gSumMag(n, rA, c) / normFactor;
        } while ((++perf->nIterations < k->maxIter &&
                  !check_convergence(perf, k->tolerance, k->relTol)) ||
                 perf->nIterations < k->minIter);
    }
    free(pA); free(pT); free(wA); free(wT); free(rA); free(rT); free(rD);
    return rc;
}

/* smoothSolver::solve (nSweeps >= 1 branch) */
static int smooth(const ffo_ldu *A, int smoother, const ffo_controls *k, double *psi,
                  const double *source, ffo_perf *perf, const ffo_comm *c)
{
    const int n = A->nCells; const size_t sz = sizeof(double) * (n + 1);
    if (smoother != FFO_GS && smoother != FFO_SYMGS) return -1;
    const int nSweeps = k->nSweeps > 0 ? k->nSweeps : 1;
    double *Apsi = (double *)malloc(sz), *tmp = (double *)malloc(sz);
    ffo_amul(A, psi, Apsi, c);
    double normFactor = ffo_norm_factor(A, psi, source, Apsi, tmp, c);
    for (int i = 0; i < n; i++) tmp[i] = source[i] - Apsi[i];
    perf->initialResidual = gSumMag(n, tmp, c) / normFactor;
    perf->finalResidual = perf->initialResidual;
    perf->nIterations = 0; perf->singular = 0;
    if (k->minIter > 0 || !check_convergence(perf, k->tolerance, k->relTol)) {
        do {
            ffo_gs_smooth(A, psi, source, nSweeps, smoother == FFO_SYMGS, c);
            ffo_residual(A, psi, source, tmp, c);
            perf->finalResidual = gSumMag(n, tmp, c) / normFactor;
        } while (((perf->nIterations += nSweeps) < k->maxIter &&
                  !check_convergence(perf, k->tolerance, k->relTol)) ||
                 perf->nIterations < k->minIter);
    }
    free(Apsi); free(tmp);
    return 0;
}

/* diagonalSolver::solve:  psi = source/diag; reports 0 iterations, residual 0 */
static int diagonal(const ffo_ldu *A, double *psi, const double *source, ffo_perf *perf)
{
    for (int i = 0; i < A->nCells; i++) psi[i] = source[i] / A->diag[i];
    perf->initialResidual = perf->finalResidual = 0.0;
    perf->nIterations = 0; perf->converged = 1; perf->singular = 0;
    return 0;
}

int ffo_solve(const ffo_ldu *A, int solver, int precond, const ffo_controls *k,
              double *psi, const double *source, ffo_perf *perf, const ffo_comm *c)
{
    memset(perf, 0, sizeof(*perf));
    switch (solver) {
    case FFO_PCG:       return A->symmetric ? pcg(A, precond, k, psi, source, perf, c) : -2;
    case FFO_PBICGSTAB: return pbicgstab(A, precond, k, psi, source, perf, c);
    case FFO_PBICG:     return pbicg(A, precond, k, psi, source, perf, c);
    case FFO_DIAGONAL:  return diagonal(A, psi, source, perf);
    case FFO_SMOOTH:    return smooth(A, precond, k, psi, source, perf, c);
    }
    return -1;
}
