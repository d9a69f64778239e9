/*
 * ffo.h -- ORACLE (test infrastructure, not product code).
 *
 * Plain-C, serial, CPU restatement of the OpenFOAM-dev linear-algebra and
 * finite-volume algorithms that fireFoam's per-time-step hot path executes
 * (reference call sites: solver/UEqn.H:3-30, solver/YEEqn.H:43-60,84-111,
 * solver/pEqn.H:3-44, solver/rhoEqn.H:33-43, solver/phrghEqn.H:43-48).
 *
 * The arithmetic itself lives in a third-party dependency that is NOT under
 * /root/reference: OpenFOAM-dev, pinned at commit
 * 940e28f63681c7e5b292096d8fd35a71acd52599 (reference CHANGELOG:1-3).  Each
 * function below names the upstream file whose published algorithm it
 * restates, and the reference call site / dictionary that selects it.
 *
 * PARITY PIN: the reference has no unit tests; its only golden data is the
 * steckler log cases/steckler/original/linux64/log.fireFoam, whose five
 * hydrostatic DICPCG solves (lines 92-101) this library reproduces digit for
 * digit (tests/test_golden_log_cpu.py): that pins Amul, sumA/normFactor, DIC
 * and PCG.  DILU, PBiCGStab, PBiCG, Gauss-Seidel and the smooth solver are
 * "parity unpinned" by reference data (oracle/README.md).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (firefoam-dev_amd/) never links it.
 *
 * Floating point: every file is compiled with -ffp-contract=off so that the
 * operation order is exactly the source order (an x86-64 OpenFOAM build has
 * no FMA contraction either).
 */
#ifndef FFO_H
#define FFO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- comm --- */
/* Stand-in for OpenFOAM Pstream: a rank-local solver sees
 * the other ranks only through these two callbacks.  NULL => serial run.   */
typedef struct ffo_comm {
    void *user;
    int rank, nRanks;
    /* sum-all-reduce of n doubles in place (gSum*, reduce(sumOp))          */
    void (*allreduce_sum)(void *user, double *vals, int n);
    /* processor-patch exchange: send[p][i] (i < size[p]) goes to the rank on
     * the other side of interface p; recv[p][i] receives that rank's send.  */
    void (*exchange)(void *user, int nIf, const int *size,
                     double *const *send, double *const *recv);
} ffo_comm;

/* ----------------------------------------------------------------- LDU --- */
/* lduAddressing + lduMatrix (upstream src/OpenFOAM/matrices/lduMatrix).     */
typedef struct ffo_ldu {
    int nCells, nFaces;
    int *l, *u;                 /* lowerAddr (owner), upperAddr (neighbour)   */
    int *ownerStart;            /* [nCells+1]                                 */
    int *losort;                /* [nFaces] faces sorted by u (stable)        */
    int *losortStart;           /* [nCells+1]                                 */
    double *diag, *upper, *lower; /* lower == upper  <=> symmetric            */
    int symmetric;
    /* coupled (processor) interfaces                                         */
    int nIf;
    int *ifSize;
    int **ifFaceCells;
    double **ifBouCoeffs;       /* interfaceBouCoeffs                         */
    double **ifIntCoeffs;       /* interfaceIntCoeffs                         */
    double **ifSend, **ifRecv;  /* scratch halo buffers                       */
    long globalCells;           /* sum of nCells over ranks (for gAverage)    */
} ffo_ldu;

ffo_ldu *ffo_ldu_create(int nCells, int nFaces, const int *l, const int *u);
void ffo_ldu_destroy(ffo_ldu *A);
/* lower == NULL => symmetric */
void ffo_ldu_set_coeffs(ffo_ldu *A, const double *diag, const double *upper,
                        const double *lower);
void ffo_ldu_set_global_cells(ffo_ldu *A, long globalCells);
void ffo_ldu_set_interfaces(ffo_ldu *A, int nIf, const int *size,
                            const int *const *faceCells,
                            const double *const *bouCoeffs,
                            const double *const *intCoeffs);

/* lduMatrix::Amul / Tmul / sumA / residual  (lduMatrixATmul.C)              */
void ffo_amul(const ffo_ldu *A, const double *x, double *y, const ffo_comm *c);
void ffo_tmul(const ffo_ldu *A, const double *x, double *y, const ffo_comm *c);
void ffo_sumA(const ffo_ldu *A, double *s);
void ffo_residual(const ffo_ldu *A, const double *x, const double *b,
                  double *r, const ffo_comm *c);
/* lduMatrix::solver::normFactor (lduMatrixSolver.C); tmp is scratch[nCells] */
double ffo_norm_factor(const ffo_ldu *A, const double *x, const double *b,
                       const double *Ax, double *tmp, const ffo_comm *c);

/* ------------------------------------------------------ preconditioners --- */
void ffo_dic_calc_rD(const ffo_ldu *A, double *rD);
void ffo_dic_precondition(const ffo_ldu *A, const double *rD, const double *r,
                          double *w);
void ffo_dilu_calc_rD(const ffo_ldu *A, double *rD);
void ffo_dilu_precondition(const ffo_ldu *A, const double *rD, const double *r,
                           double *w);
void ffo_dilu_preconditionT(const ffo_ldu *A, const double *rD,
                            const double *r, double *w);
/* GaussSeidelSmoother / symGaussSeidelSmoother ::smooth                     */
void ffo_gs_smooth(const ffo_ldu *A, double *psi, const double *b, int nSweeps,
                   int symmetric_sweep, const ffo_comm *c);

/* -------------------------------------------------------------- solvers --- */
enum { FFO_PCG = 0, FFO_PBICGSTAB = 1, FFO_PBICG = 2, FFO_DIAGONAL = 3,
       FFO_SMOOTH = 4 };
enum { FFO_NONE = 0, FFO_DIC = 1, FFO_DILU = 2, FFO_GS = 3, FFO_SYMGS = 4,
       FFO_DIAGONALP = 5 };

typedef struct ffo_perf {
    double initialResidual, finalResidual;
    int nIterations, converged, singular;
} ffo_perf;

typedef struct ffo_controls {
    double tolerance, relTol;
    int minIter, maxIter, nSweeps;
} ffo_controls;

int ffo_solve(const ffo_ldu *A, int solver, int precond, const ffo_controls *k,
              double *psi, const double *source, ffo_perf *perf,
              const ffo_comm *c);

/* ----------------------------------------------------------- synthetic --- */
/* u(seed,idx) = (splitmix64(seed ^ idx) >> 11) * 2^-53         */
double ffo_hash_u(uint64_t seed, uint64_t idx);

/* ------------------------------------------------------------- hex mesh --- */
/* blockMesh single-block numbering                             */
void ffo_hex_counts(int nx, int ny, int nz, long *nCells, long *nFaces);
void ffo_hex_ldu(int nx, int ny, int nz, int *l, int *u);

/* in-process multi-domain runner: P pthreads, one ffo_ldu each, block-Jacobi
 * preconditioning, pthread-barrier halo exchange + reductions              */
int ffo_solve_multi(int P, ffo_ldu **A, const int *const *ifNbrRank,
                    const int *const *ifNbrPatch, int solver, int precond,
                    const ffo_controls *k, double **psi,
                    const double *const *source, ffo_perf *perf);

#ifdef __cplusplus
}
#endif
#endif
