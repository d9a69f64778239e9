/*
 * ffm.h -- C ABI of libffm.so, the MI355X (gfx950) implementation of fireFoam's
 * per-time-step hot path: lduMatrix kernels, preconditioned Krylov solves and
 * finite-volume assembly, as hand-written HIP kernels.
 *
 * This is the drop-in boundary.  Each entry point names the
 * reference call site it serves and the OpenFOAM-dev (pinned @940e28f,
 * reference CHANGELOG:1-3; not vendored in the reference) interface it
 * replaces.  INTEGRATION.md shows the OpenFOAM-side binding
 * (a lduMatrix::solver registered through the run-time selection table and
 * loaded with controlDict `libs (...)`, the mechanism the reference itself
 * uses in cases/pyrolysis1D/system/controlDict:59-62).
 *
 * Conventions: plain C; opaque handles; every function returns 0 on success or
 * a negative ffm_status; no exceptions cross the boundary; pointers are HOST
 * pointers unless the parameter name ends in `_d` (device pointer in the
 * context's HIP device); the library never frees caller memory; one context =
 * one GPU = one host thread (OpenFOAM runs one rank per process).
 * All floating point is fp64, all labels int32 (OpenFOAM WM_PRECISION_OPTION=DP,
 * WM_LABEL_SIZE=32).
 */
#ifndef FFM_H
#define FFM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ffm_ctx ffm_ctx;
typedef struct ffm_ldu ffm_ldu;

typedef enum ffm_status {
    FFM_OK = 0,
    FFM_ERR_ARG = -1,        /* bad argument / inconsistent sizes             */
    FFM_ERR_ADDR = -2,       /* LDU addressing not upper-triangular ordered   */
    FFM_ERR_HIP = -3,        /* HIP runtime error (see ffm_last_error)        */
    FFM_ERR_NODEVICE = -4,   /* no usable GPU                                 */
    FFM_ERR_UNSUPPORTED = -5,/* solver/preconditioner combination not offered */
    FFM_ERR_COMM = -6        /* RCCL error                                    */
} ffm_status;

/* lduMatrix::solver run-time table keys (fvSolution `solver`), reference
 * cases/steckler/system/fvSolution:21-81, cases/wallFireSpread2D/system/fvSolution:115-152 */
typedef enum ffm_solver {
    FFM_PCG = 0, FFM_PBICGSTAB = 1, FFM_PBICG = 2, FFM_DIAGONAL = 3, FFM_SMOOTH = 4,
    FFM_GAMG = 5      /* through ffm_gamg_* only (an agglomeration object per mesh); ffm_solve_d refuses it */
} ffm_solver;
/* fvSolution `preconditioner` / `smoother` */
typedef enum ffm_precond {
    FFM_NONE = 0, FFM_DIC = 1, FFM_DILU = 2, FFM_GS = 3, FFM_SYMGS = 4, FFM_DIAGONALP = 5
} ffm_precond;

/* SolverPerformance<scalar> (what `DICPCG:  Solving for p_rgh, Initial residual
 * = ..., Final residual = ..., No Iterations n` prints; golden log
 * cases/steckler/original/linux64/log.fireFoam:220)                          */
typedef struct ffm_perf {
    double initialResidual, finalResidual;
    int nIterations, converged, singular;
} ffm_perf;

/* ------------------------------------------------------------------ context */
/* stream: a hipStream_t to launch on, or NULL to create a private stream.    */
int ffm_ctx_create(int device, void *stream, ffm_ctx **out);
int ffm_ctx_destroy(ffm_ctx *ctx);
int ffm_ctx_sync(ffm_ctx *ctx);
void *ffm_ctx_stream(ffm_ctx *ctx);
const char *ffm_last_error(void);
const char *ffm_version(void);

/* device memory helpers so a host without HIP headers (the C++ Foam layer, an
 * OpenFOAM shim) can hold fields on the GPU                                  */
int ffm_malloc(ffm_ctx *ctx, size_t bytes, void **ptr_d);
int ffm_free(ffm_ctx *ctx, void *ptr_d);
/* the same allocator without the zero-fill: for a block whose every byte the caller overwrites next (the result of a field
 * operation, a copy) -- on small meshes the fill kernel costs as much as the operation */
int ffm_malloc_uninit(ffm_ctx *ctx, size_t bytes, void **ptr_d);
/* ffm_malloc / ffm_free go through a stream-ordered caching allocator (freed blocks are reused by later allocations of the same
 * size class on the context stream, without synchronising; every block is handed out zero-filled); ffm_ctx_trim returns the cached
 * blocks to the runtime */
int ffm_ctx_trim(ffm_ctx *ctx);
int ffm_memcpy_h2d(ffm_ctx *ctx, void *dst_d, const void *src, size_t bytes);
int ffm_memcpy_d2h(ffm_ctx *ctx, void *dst, const void *src_d, size_t bytes);
int ffm_memcpy_d2d(ffm_ctx *ctx, void *dst_d, const void *src_d, size_t bytes);
int ffm_memset(ffm_ctx *ctx, void *dst_d, int value, size_t bytes);

/* element-wise field algebra (Field<scalar> operators of the Foam layer): one pass per operator, like
 * OpenFOAM's own tmp-field evaluation, so an expression rounds the way the reference's does                        */
enum { FFM_OP_ADD = 0, FFM_OP_SUB = 1, FFM_OP_MUL = 2, FFM_OP_DIV = 3, FFM_OP_MAX = 4, FFM_OP_MIN = 5,
       FFM_OP_NEGSEL = 6 /* a < 0 ? b : a  (a marker value selects the other operand) */ };
enum { FFM_UN_NEG = 0, FFM_UN_SQR = 1, FFM_UN_MAG = 2, FFM_UN_SQRT = 3, FFM_UN_POS0 = 4 /* pos0(x) = x >= 0 ? 1 : 0 */ };
int ffm_field_binary(ffm_ctx *ctx, int op, long n, const double *a_d, const double *b_d, double *out_d);
/* out = a op s, or s op a when scalarFirst != 0 */
int ffm_field_scalar(ffm_ctx *ctx, int op, long n, const double *a_d, double s, int scalarFirst, double *out_d);
int ffm_field_unary(ffm_ctx *ctx, int op, long n, const double *a_d, double *out_d);
int ffm_field_fill(ffm_ctx *ctx, long n, double s, double *out_d);
/* One element-wise expression over fields in ONE pass (the Foam layer's lazily evaluated field algebra, include/ffmFoam.H: dField):
 * a postfix program over an operand stack of depth <= 4.  code[k] = kind << 12 | arg:
 *   FFM_EVAL_LOAD   arg = index into arrays[]          push arrays[arg][i]
 *   FFM_EVAL_IMM    arg = index into imm[]             push imm[arg]
 *   FFM_EVAL_BINARY arg = FFM_OP_*                     b = pop, a = pop, push (a op b)
 *   FFM_EVAL_UNARY  arg = FFM_UN_*                     top = op(top)
 * The value left on the stack is out[i]; at least one array (a constant is ffm_field_fill).  Every operation rounds as the separate ffm_field_binary / _scalar / _unary call does, so the
 * result is bit for bit that of the chain of calls (OpenFOAM's tmp<Field> operator chains, e.g. solver/pEqn.H:5-12); `out` must not
 * overlap an input.                                                                                                              */
enum { FFM_EVAL_LOAD = 1, FFM_EVAL_IMM = 2, FFM_EVAL_BINARY = 3, FFM_EVAL_UNARY = 4 };
#define FFM_EVAL_MAX_ARRAYS 8
#define FFM_EVAL_MAX_IMM 8
#define FFM_EVAL_MAX_INSTR 32
int ffm_field_eval(ffm_ctx *ctx, long n, int nArrays, const double *const *arrays_d, int nImm, const double *imm, int nInstr,
                   const unsigned short *code, double *out_d);

/* ----------------------------------------------------------- lduAddressing */
/* Replaces lduAddressing + lduMatrix construction
 * [upstream src/OpenFOAM/matrices/lduMatrix/lduAddressing/lduAddressing.H].
 * lowerAddr/upperAddr: owner/neighbour cell of every internal face in
 * OpenFOAM's upper-triangular order (l<u, faces sorted by owner).  Builds on
 * the device: ownerStart, losort, losortStart, the dependency levels of the
 * DIC/DILU/Gauss-Seidel sweeps and a level-major cell renumbering (skipped
 * when the caller's numbering is already level-major).                       */
int ffm_ldu_create(ffm_ctx *ctx, int nCells, int nFaces, const int *lowerAddr,
                   const int *upperAddr, ffm_ldu **out);
/* Decomposed (one rank per GPU) form: cells [0,nOwned) are this rank's, cells
 * [nOwned, nOwned+nGhost) are copies of neighbour-rank cells (one layer across the
 * cut faces, which are ordinary faces owned by the owned cell).  Rows exist for owned
 * cells only; DIC/DILU ignore faces towards ghosts (block-Jacobi, as OpenFOAM's
 * processor interfaces); Amul refreshes the ghost entries of its argument first
 * (ffm_ldu_set_ghost_exchange).  Cell fields have nOwned+nGhost entries.          */
int ffm_ldu_create_ext(ffm_ctx *ctx, int nOwned, int nGhost, int nFaces, const int *lowerAddr,
                       const int *upperAddr, ffm_ldu **out);
int ffm_renumber_levels_ext(int nOwned, int nGhost, int nFaces, const int *lowerAddr,
                            const int *upperAddr, int *newToOldCell, int *newToOldFace);
/* Same with a grouping hint for the sweeps: groupHint[c] = any label shared by owned cells that should be swept by one
 * workgroup (e.g. a 2-D tile of cell columns, from the cell centres).  Used when the label classes form an acyclic
 * dependency graph, ignored otherwise (NULL: the library chunks the caller's cell order).                         */
int ffm_ldu_create_hint(ffm_ctx *ctx, int nOwned, int nGhost, int nFaces, const int *lowerAddr,
                        const int *upperAddr, const int *groupHint, ffm_ldu **out);
int ffm_renumber_hint(int nOwned, int nGhost, int nFaces, const int *lowerAddr, const int *upperAddr,
                      const int *groupHint, int *newToOldCell, int *newToOldFace);
/* a group hint from the cell centres C[3][nCells] (mesh.C()): 2-D tiles, about tileCells (0: default 16) cells wide, of
 * columns along the axis consecutive cell labels follow; pure host code                                              */
int ffm_tile_hint_from_centres(int nCells, const double *C, int tileCells, int *hint);
/* neighbour q (rank nbrRank[q]) receives this rank's cells sendCells[...] (sendCount[q] of
 * them, concatenated) and fills recvCount[q] consecutive ghost cells, in neighbour order    */
int ffm_ldu_set_ghost_exchange(ffm_ldu *ldu, int nNbr, const int *nbrRank, const int *sendCount,
                               const int *sendCells, const int *recvCount);
int ffm_halo_refresh_d(ffm_ldu *ldu, double *field_d);
int ffm_ldu_nowned(const ffm_ldu *ldu);
int ffm_ldu_destroy(ffm_ldu *ldu);
int ffm_ldu_ncells(const ffm_ldu *ldu);
int ffm_ldu_nfaces(const ffm_ldu *ldu);
int ffm_ldu_nlevels(const ffm_ldu *ldu);
/* which sweep kernels DIC / DILU / Gauss-Seidel use on this matrix: 0 one launch per dependency level, 2 tiled wavefront
 * (ffm_tile.hip: group hint or detected blockMesh box)                                                              */
int ffm_ldu_sweep_mode(const ffm_ldu *ldu);
/* 1 when the caller's cell numbering is used as is (no permutation passes)   */
int ffm_ldu_is_native_order(const ffm_ldu *ldu);
/* new->old cell permutation chosen by the library (host, int[nCells])        */
int ffm_ldu_get_cell_order(const ffm_ldu *ldu, int *newToOld);

/* Level-major renumbering of an LDU graph, for hosts that want to renumber the
 * mesh once (like renumberMesh) so that no permutation pass is ever needed:
 * newToOldCell[nCells], newToOldFace[nFaces]; the renumbered faces are again in
 * upper-triangular order and keep owner<neighbour.  Pure host code.          */
int ffm_renumber_levels(int nCells, int nFaces, const int *lowerAddr,
                        const int *upperAddr, int *newToOldCell,
                        int *newToOldFace);

/* lduMatrix::diag()/upper()/lower(): lower == NULL => symmetric matrix.      */
int ffm_ldu_set_coeffs(ffm_ldu *ldu, const double *diag, const double *upper,
                       const double *lower);
int ffm_ldu_set_coeffs_d(ffm_ldu *ldu, const double *diag_d,
                         const double *upper_d, const double *lower_d);

/* Native face layout (for hosts that assemble on the device in the library's own
 * layout and skip the LDU-order gather): the library stores faces as a sliced
 * owner-ELL; nNative >= nFaces entries incl. padding.  callerToNative[f] = native
 * index of caller face f.  Only valid when ffm_ldu_is_native_order() == 1.     */
int ffm_ldu_n_native_faces(const ffm_ldu *ldu);
int ffm_ldu_get_face_map(const ffm_ldu *ldu, int *callerToNative);
int ffm_ldu_set_coeffs_native_d(ffm_ldu *ldu, const double *diag_d,
                                const double *upper_d, const double *lower_d);
/* zero-copy form of the same: the matrix reads the caller's device arrays until the next set/bind call (the caller keeps them
 * alive and unchanged); offDiagUnchanged != 0 tells the library that upper/lower hold the same values as at the previous call
 * (the components of a vector equation differ in the boundary diagonal only: fvMatrix::solveSegregated)               */
int ffm_ldu_bind_coeffs_native_d(ffm_ldu *ldu, const double *diag_d, const double *upper_d,
                                 const double *lower_d, int offDiagUnchanged);
/* ends a bind: the matrix refers to its own buffers again (contents unspecified until the next set / bind), the caller may free its arrays */
int ffm_ldu_unbind_coeffs(ffm_ldu *ldu);

/* processor patches (lduInterface / interfaceBouCoeffs / interfaceIntCoeffs):
 * face i of patch p couples cell faceCells[p][i] (caller numbering) with face i
 * of the matching patch on rank neighbRank[p]; as in OpenFOAM there is one
 * patch per neighbour rank (or both sides list their common patches in the
 * same order).  Amul: y[faceCells] -= bouCoeffs*psi_neighbour.  DIC/DILU ignore
 * the interfaces (block-Jacobi).  Needs ffm_comm_init / ffm_comm_init_host.   */
int ffm_ldu_set_interfaces(ffm_ldu *ldu, int nPatches, const int *patchSizes,
                           const int *const *faceCells,
                           const double *const *bouCoeffs,
                           const double *const *intCoeffs,
                           const int *neighbRank);
int ffm_ldu_set_global_cells(ffm_ldu *ldu, long globalCells);
/* Pair tags of the point-to-point messages of one exchange (kind 0: the processor patches of ffm_ldu_set_interfaces, kind 1:
 * the neighbour entries of ffm_ldu_set_ghost_exchange): tags[i] is a label both ranks give to the two entries that exchange
 * with each other.  Messages are posted in ascending tag order, so two ranks that share several patches (cyclic / periodic
 * two-rank decompositions, split processor patches) match them correctly whatever their local patch order.  Without tags
 * the entry order is used, i.e. both sides must list their common patches in the same order (as OpenFOAM does).          */
int ffm_ldu_set_exchange_tags(ffm_ldu *ldu, int kind, int n, const int *tags);

/* ------------------------------------------------------- lduMatrix kernels */
/* lduMatrix::Amul -- the metric kernel (solver/pEqn.H:39 via PCG; UEqn.H() in
 * solver/pEqn.H:5).  x_d, y_d in the caller's cell numbering.                */
int ffm_spmv(ffm_ldu *ldu, const double *x_d, double *y_d);
int ffm_tmul(ffm_ldu *ldu, const double *x_d, double *y_d);      /* ::Tmul     */
int ffm_sumA(ffm_ldu *ldu, double *s_d);                         /* ::sumA     */
int ffm_residual(ffm_ldu *ldu, const double *x_d, const double *b_d, double *r_d);
/* preconditioner pieces, exposed for parity tests             */
int ffm_precond_setup(ffm_ldu *ldu, int precond, double *rD_out_d /*nullable*/);
int ffm_precond_apply(ffm_ldu *ldu, int precond, int transpose,
                      const double *r_d, double *w_d);
int ffm_gs_smooth(ffm_ldu *ldu, int symmetric_sweep, int nSweeps, double *psi_d,
                  const double *b_d);

/* lduMatrix::solver::New(...)->solve(psi, source):
 * psi in/out, source in; controls = fvSolution entries tolerance, relTol,
 * minIter, maxIter, nSweeps (defaults 1e-6, 0, 0, 1000, 1).                  */
int ffm_solve_d(ffm_ldu *ldu, int solver, int precond, double tolerance,
                double relTol, int minIter, int maxIter, int nSweeps,
                double *psi_d, const double *source_d, ffm_perf *out);
int ffm_solve(ffm_ldu *ldu, int solver, int precond, double tolerance,
              double relTol, int minIter, int maxIter, int nSweeps,
              double *psi, const double *source, ffm_perf *out);
/* nSys systems that share the off-diagonal coefficients and differ in the diagonal and the right-hand side: the components of a
 * vector equation (fvMatrix<Type>::solveSegregated, OpenFOAM-dev fvMatrixSolve.C; solver/UEqn.H:19-30) and the species equations
 * under a multivariateSelection scheme with one diffusivity (solver/YEEqn.H:43-60).  Every system gets the solve a call of
 * ffm_ldu_bind_coeffs_native_d(diag_d[i], upper_d, lower_d) + ffm_solve_d would give it -- the same operations in the same order,
 * its own iteration count and residuals in out[i] -- but for PBiCGStab with DILU / DIC on a tiled matrix in the library's cell order
 * the systems (at most 4 at a time) advance in lock step and the preconditioner sweeps and calcReciprocalD of those still iterating
 * are one sweep each (coefficients and addressing read once).  Any other selection: one solve after the other.
 * diag_d[i] [cells], upper_d / lower_d [native faces; lower_d NULL or == upper_d: symmetric], psi_d[i] in/out, source_d[i].
 * The matrix is left bound to the caller's arrays as by ffm_ldu_bind_coeffs_native_d.                                         */
int ffm_solve_multi_d(ffm_ldu *ldu, int nSys, int solver, int precond, double tolerance, double relTol, int minIter, int maxIter,
                      const double *const *diag_d, const double *upper_d, const double *lower_d, double *const *psi_d,
                      const double *const *source_d, ffm_perf *out /* [nSys] */);

/* timing helper for bench.py: runs `reps` back-to-back launches of the SpMV
 * kernel used inside the solvers on the context stream, bracketed by HIP
 * events on that stream; returns the average kernel time in milliseconds.    */
int ffm_bench_spmv(ffm_ldu *ldu, const double *x_d, double *y_d, int reps,
                   double *avg_ms);
/* the same for one preconditioner application (DIC: forward + backward sweep) on the bound matrix */
int ffm_bench_precond(ffm_ldu *ldu, int precond, const double *r_d, double *w_d, int reps, double *avg_ms);

/* ------------------------------------------------------- finite-volume mesh */
/* fvMesh geometry on the device (mesh.V(), Sf(), magSf(), weights(), deltaCoeffs(),
 * boundary(); reference use: solver/UEqn.H:28, solver/pEqn.H:7).  The LDU must
 * be in the library's cell order (ffm_renumber_levels).  Host arrays: V[N],
 * C[3][N]; Sf[3][F], magSf[F], weights[F], deltaCoeffs[F] in LDU face order;
 * per patch faceCells[n], Sf[3][n], deltaCoeffs[n].  Face fields handed to the
 * fv entry points are in the NATIVE face layout (ffm_mesh_nnative entries,
 * conversion helpers below); boundary fields have ffm_mesh_nboundary entries,
 * patches concatenated in the order given here.                              */
typedef struct ffm_mesh ffm_mesh;
int ffm_mesh_create(ffm_ldu *ldu, const double *V, const double *C,
                    const double *Sf, const double *magSf, const double *weights,
                    const double *deltaCoeffs, int nPatches, const int *patchSizes,
                    const int *const *faceCells, const double *const *patchSf,
                    const double *const *patchDeltaCoeffs, ffm_mesh **out);
/* mesh.Cf() of the internal faces, host array Cf[3][F] in LDU face order (needed by the LUST correction only) */
int ffm_mesh_set_face_centres(ffm_mesh *mesh, const double *Cf);
/* mesh.nonOrthCorrectionVectors() of the internal faces, host array [3][F] in LDU face order; needed by the `corrected`
 * snGrad / laplacian schemes on non-orthogonal meshes only (cases/wallFireSpread2D/system/fvSchemes: `Gauss linear corrected`) */
int ffm_mesh_set_nonorth_correction(ffm_mesh *mesh, const double *corrVec);
int ffm_mesh_destroy(ffm_mesh *mesh);
int ffm_mesh_nboundary(const ffm_mesh *mesh);
/* device geometry fields for the Foam layer: 0 V[N], 1 magSf[nNative], 2 deltaCoeffs[nNative], 3 weights[nNative],
 * 4 boundary magSf[B], 5 boundary deltaCoeffs[B], 6-8 boundary Sf x,y,z [B], 9-11 Sf x,y,z of the internal faces [nNative],
 * 12-14 cell centres x,y,z [N]                                                                                 */
const double *ffm_mesh_geometry_d(const ffm_mesh *mesh, int which);
int ffm_mesh_nnative(const ffm_mesh *mesh);
int ffm_faces_to_native(const ffm_mesh *mesh, const double *lduOrder, double *native_d);
int ffm_faces_from_native(const ffm_mesh *mesh, const double *native_d, double *lduOrder);

/* ------------------------------------------------------- fvc:: (explicit)    */
/* solver/UEqn.H:23-29, solver/pEqn.H:4-17,43-44, solver/YEEqn.H:87-95.  All
 * pointers are device pointers; *_f = face field (native), *_b = boundary field. */
int ffm_fvc_interpolate(ffm_mesh *m, const double *w_f, const double *vf, double *out_f);
int ffm_fvc_snGrad(ffm_mesh *m, const double *vf, double *out_f);
int ffm_fvc_snGrad_b(ffm_mesh *m, const double *vf, const double *vb, double *out_b);
int ffm_fvc_flux(ffm_mesh *m, const double *vx, const double *vy, const double *vz, double *out_f);
int ffm_fvc_surface_integrate(ffm_mesh *m, const double *ssf_f, const double *ssf_b, double *out);
int ffm_fvc_surface_sum(ffm_mesh *m, const double *ssf_f, const double *ssf_b, double *out);
int ffm_fvc_grad(ffm_mesh *m, const double *vf, const double *vb, double *gx, double *gy, double *gz);
int ffm_fvc_reconstruct(ffm_mesh *m, const double *ssf_f, const double *ssf_b, double *ox, double *oy, double *oz);
/* correctedSnGrad::correction(vf) = nonOrthCorrectionVectors & interpolate(grad(vf)), internal faces (one scalar component).
 * `corrected` snGrad = ffm_fvc_snGrad + this; `Gauss linear corrected` laplacian = the uncorrected matrix with
 * source -= V*ffm_fvc_surface_integrate(gamma_f*magSf*this, 0) */
int ffm_fvc_snGrad_correction(ffm_mesh *m, const double *gx, const double *gy, const double *gz, double *out_f);
/* limitedSurfaceInterpolationScheme weights: scheme 0 upwind, 1 linear,
 * 2 limitedLinear k, 3 limitedLinear01 k with bounds [lo,hi], 4 LUST (0.75 linear +
 * 0.25 upwind; vf and gradients unused), 5 linearUpwind (upwind weights; its explicit correction is
 * ffm_fv_linear_upwind_correction)  (cases/steckler/system/fvSchemes:28-54) */
int ffm_fv_limited_weights(ffm_mesh *m, int scheme, double k, double lo, double hi,
                           const double *phi_f, const double *vf, const double *gx,
                           const double *gy, const double *gz, double *out_w_f);
/* multivariateSelectionScheme (solver/YEEqn.H:1-10 `fv::convectionScheme<scalar>::New(mesh, fields, phi, mesh.divScheme("div(phi,Yi_h)"))`,
 * cases/steckler/system/fvSchemes:36-47): ONE limiter for all fields of the table, the face-wise minimum of the member schemes' limiters,
 * and the weights made of it.  ffm_fv_limited_limiter: the limiter of limitedLinear k (scheme 2) / limitedLinear01 k (scheme 3) of one
 * field into lim_f, or the running minimum when accumulateMin != 0; ffm_fv_weights_from_limiter: limiter*linear + (1 - limiter)*upwind. */
int ffm_fv_limited_limiter(ffm_mesh *m, int scheme, double k, double lo, double hi, const double *phi_f, const double *vf,
                           const double *gx, const double *gy, const double *gz, double *lim_f, int accumulateMin);
int ffm_fv_weights_from_limiter(ffm_mesh *m, const double *phi_f, const double *lim_f, double *out_w);
/* the same in one pass: the common weights over nf fields (schemes[i]: 2 limitedLinear, 3 limitedLinear01), bitwise equal to the
 * running-minimum chain above */
int ffm_fv_multivariate_weights(ffm_mesh *m, int nf, const int *schemes, double k, double lo, double hi, const double *phi_f,
                                const double *const *vf, const double *const *gx, const double *const *gy, const double *const *gz,
                                double *out_w);
/* ffm_fvc_grad_multi over the nf <= 6 fields + ffm_fv_multivariate_weights in ONE pass with the cell values staged through LDS on the tile
 * numbering (a workgroup walks a run of dependency levels of one tile; the values of the levels e-1, e, e+1 sit in an LDS window, a cell
 * forms its own gradients in registers, a face is written by its upwind cell): bit for bit the two-pass form without the 3 nf gradient
 * arrays.  vb: the fields' boundary values.  FFM_ERR_UNSUPPORTED where the matrix has no tile plan, the mesh is one block of a decomposed
 * case, cells have more than 3 lower / upper neighbours or nf > 6: the caller then runs the two-pass form.                            */
int ffm_fv_multivariate_weights_tiled(ffm_mesh *m, int nf, const int *schemes, double k, double lo, double hi, const double *phi_f,
                                      const double *const *vf, const double *const *vb, double *out_w);

/* filteredLinear2V k l: the face weights of a VECTOR field, one limiter per face for its three components
 * (`div(phi,U) Gauss filteredLinear2V 0.2 0.05`, cases/wallFireSpread2D/system/fvSchemes:41, cases/pyrolysis1D/system/fvSchemes:39;
 * solver/UEqn.H:5).  U[3], gx[3], gy[3], gz[3]: host arrays of device pointers, g?[c] = the gradient of component c.        */
int ffm_fv_filtered_linear2V_weights(ffm_mesh *m, double k, double l, const double *phi_f, const double *const *U,
                                     const double *const *gx, const double *const *gy, const double *const *gz, double *out_w_f);

/* LUST<Type>::correction(vf) for one scalar component, internal faces: 0.25*(Cf - C_c) & grad(vf)_c with c the upwind cell
 * (`div(phi,U) Gauss LUST grad(U)`, cases/steckler/system/fvSchemes:32; solver/UEqn.H:5).  gaussConvectionScheme::fvmDiv
 * then adds fvc::surfaceIntegrate(phi*correction) to the matrix: source -= V * ffm_fvc_surface_integrate(phi_f*corr_f, 0). */
int ffm_fv_lust_correction(ffm_mesh *m, const double *phi_f, const double *gx, const double *gy,
                           const double *gz, double *out_f);
/* linearUpwind<Type>::correction(vf): (Cf - C_c) & grad(vf)_c with c the upwind cell, no factor
 * (`div(Ji,Ii_h) Gauss linearUpwind grad(Ii_h)`, cases/wallFireSpread2D/system/fvSchemes:58); used like the LUST one. */
int ffm_fv_linear_upwind_correction(ffm_mesh *m, const double *phi_f, const double *gx, const double *gy,
                                    const double *gz, double *out_f);

/* ------------------------------------------------------- fvm:: (implicit)    */
/* [fvm::ddt(rho,.)] + [fvm::div(phi,.)] (+/-) [fvm::laplacian(gamma,.)] in one
 * pass; absent terms: NULL.  Writes lduMatrix diag/upper/lower (native layout).
 * upper = lower = NULL with a ddt term only (it has no face coefficients); lower = NULL without convection (symmetric: lower = upper). */
int ffm_fvm_transport(ffm_mesh *m, double rDeltaT, const double *rho,
                      const double *phi_f, const double *w_f, const double *gamma_f,
                      int laplacianSign, double *diag, double *upper, double *lower);
/* internalCoeffs / boundaryCoeffs of the same terms for a patch field in
 * `mixed` form (valueFraction f, refValue, refGradient); fixedValue,
 * zeroGradient, fixedGradient, inletOutlet are special cases               */
int ffm_fvm_boundary_coeffs(ffm_mesh *m, const double *phi_b, const double *gamma_b,
                            int laplacianSign, const double *f, const double *ref,
                            const double *refGrad, double *internalCoeffs, double *boundaryCoeffs);
int ffm_bc_values(ffm_mesh *m, const double *f, const double *ref, const double *refGrad,
                  const double *vf, double *out_b);
/* fvMatrix::addBoundaryDiag + addBoundarySource (+ V*su)                     */
int ffm_fvm_add_boundary(ffm_mesh *m, const double *internalCoeffs, const double *boundaryCoeffs,
                         const double *diag, const double *source, const double *su,
                         double *diagOut, double *sourceOut);
/* fvMatrix<Type>::relax(alpha) (solver/UEqn.H:13, solver/YEEqn.H:56,107; equation factors e.g.
 * cases/wallFireSpread2D/system/fvSolution:194-200): diagonal dominance + under-relaxation in place,
 * source_c += (D - D0)*psiPrev_c for nCmpt = 1 or 3 components sharing the coefficients.            */
int ffm_fvm_relax(ffm_mesh *m, double alpha, int nCmpt, const double *upper, const double *lower,
                  const double *ic0, const double *ic1, const double *ic2, double *diag,
                  const double *psiPrev0, const double *psiPrev1, const double *psiPrev2,
                  double *source0, double *source1, double *source2);
/* fvMatrix::A(), H() (one component), flux()  (solver/pEqn.H:3,5,43)          */
int ffm_fvm_A(ffm_mesh *m, int nCmpt, const double *diag, const double *ic0, const double *ic1,
              const double *ic2, double *out);
int ffm_fvm_H(ffm_mesh *m, int nCmpt, int cmpt, const double *upper, const double *lower,
              const double *source, const double *ic0, const double *ic1, const double *ic2,
              const double *bcCmpt, const double *psi, double *out);
/* the three components of fvMatrix<vector>::H() in one pass, times rAU when given: HbyA = rAU*UEqn.H() (solver/pEqn.H:9);
 * source / internalCoeffs / boundaryCoeffs / psi / out: arrays of three device pointers; bitwise equal to three ffm_fvm_H
 * calls followed by the product */
int ffm_fvm_HbyA3(ffm_mesh *m, const double *upper, const double *lower, const double *const *source,
                  const double *const *internalCoeffs, const double *const *boundaryCoeffs, const double *const *psi,
                  const double *rAU, double *const *out);
/* p_rghEqn of solver/pEqn.H:28-36 in one pass: fvm::ddt(psi, p_rgh) + fvc::ddt(psi, rho)*gh + fvc::ddt(psi)*pRef +
 * fvc::div(phiHbyA) - fvm::laplacian(gamma, p_rgh) with its boundary coefficients (internalCoeffs / boundaryCoeffs from
 * ffm_fvm_boundary_coeffs) already added: upper / lower [native faces], diagOut / sourceOut [cells] are what the solver takes.
 * Bitwise equal to ffm_fvm_transport + ffm_fvc_surface_integrate + the three source updates + ffm_fvm_add_boundary. */
int ffm_fvm_pressure_eqn(ffm_mesh *m, double rDeltaT, const double *psi, const double *psi0, const double *p0,
                         const double *rho, const double *rho0, const double *gh, double pRef, const double *gamma_f,
                         const double *phiHbyA_f, const double *phiHbyA_b, const double *internalCoeffs,
                         const double *boundaryCoeffs, double *upper, double *lower, double *diagOut, double *sourceOut);
/* three face passes of the pressure corrector (solver/pEqn.H:9-19,43-44), each two per-operator passes in one with their arithmetic:
 * phig = -rhorAUf*ghf*fvc::snGrad(rho)*magSf;  phiHbyA = (fvc::flux(rho*HbyA) + rhorAUf*ddtCorr) + phig;
 * fl = p_rghEqn.flux(), phi = phiHbyA + fl, t = (fl + phig)/rhorAUf (internal faces, native layout; symmetric matrix: lower = upper) */
int ffm_pc_phig(ffm_mesh *m, const double *rhorAUf, const double *ghf, const double *rho, double *phig);
int ffm_pc_phiHbyA(ffm_mesh *m, const double *rho, const double *vx, const double *vy, const double *vz, const double *rhorAUf,
                   const double *ddtCorr, const double *phig, double *phiHbyA);
int ffm_pc_flux(ffm_mesh *m, const double *upper, const double *lower, const double *psi, const double *phiHbyA, const double *phig,
                const double *rhorAUf, double *flux_f, double *phi_f, double *t_f);
/* solver/UEqn.H:23-29: the face flux of fvc::reconstruct, t = (-ghf*fvc::snGrad(rho) - fvc::snGrad(p_rgh))*magSf, in one pass */
int ffm_ue_buoyancy_flux(ffm_mesh *m, const double *ghf, const double *rho, const double *p_rgh, double *t_f);
/* fvc::flux(rho*v) on the internal faces (solver/pEqn.H:15 fvc::flux(rho*HbyA); the old-time flux of fvc::ddtCorr) without
 * storing the product fields; bitwise equal to ffm_fvc_flux of the products */
int ffm_fvc_flux_rho(ffm_mesh *m, const double *rho, const double *vx, const double *vy, const double *vz, double *out_f);
int ffm_fvm_flux(ffm_mesh *m, const double *upper, const double *lower, const double *internalCoeffs,
                 const double *boundaryCoeffs, const double *psi, double *out_f, double *out_b);

/* ------------------------------------------------------- fused assembly passes */
/* The entry points above evaluate one operator per pass, like OpenFOAM's tmp-field algebra (and like the Foam layer of
 * include/ffmFoam.H calls them).  These produce the same numbers, bit for bit, in one pass per equation and for up to 4
 * fields that share the flux / density / diffusivity per launch (the species loop of solver/YEEqn.H:37-67); the compiled
 * time step (ffm_plume_step) is built on them.  Arrays of nf device pointers are HOST arrays.                          */
/* fvc::grad (Gauss linear) of nf <= 4 fields */
int ffm_fvc_grad_multi(ffm_mesh *m, int nf, const double *const *vf, const double *const *vb, double *const *gx,
                       double *const *gy, double *const *gz);
/* per field i: fvm::ddt(rho,vf_i) + fvm::div(phi,vf_i) [scheme 2 limitedLinear k | 3 limitedLinear01 k in [lo,hi], limiter from
 * vf_i and its gradient] - fvm::laplacian(gamma,vf_i) == su_i + su2_i - fvm::Sp(sp_i, vf_i) (each nullable: a NULL array or a NULL
 * entry), minus the explicit volume terms expl3[3*i+0..2] (all three or none; NULL array: none), with the boundary coefficients
 * of the mixed condition (f_i, ref_i, refGrad_i) added:
 * diag_i, upper_i, lower_i, source_i are what fvMatrix::solveSegregated hands to the linear solver                       */
int ffm_fvm_scalar_transport_multi(ffm_mesh *m, int nf, int scheme, double k, double lo, double hi, double rDeltaT,
                                   const double *rho, const double *rho0, const double *phi_f, const double *phi_b,
                                   const double *gamma_f, const double *gamma_b,
                                   const double *const *vf, const double *const *gx, const double *const *gy,
                                   const double *const *gz, const double *const *vf0, const double *const *f,
                                   const double *const *ref, const double *const *refGrad, const double *const *su,
                                   const double *const *su2, const double *const *sp,
                                   const double *const *expl3, double *const *diag, double *const *upper,
                                   double *const *lower, double *const *source);
/* the same pass with the face weights given (a multivariateSelection scheme's common weights: ffm_fv_multivariate_weights) instead of
 * one limiter per field -- what mvConvection->fvmDiv(phi, Yi) of solver/YEEqn.H:44 assembles.  With common weights and one gamma every
 * field has the SAME off-diagonal coefficients: upper[i] = lower[i] = NULL for the fields i >= n0 (n0 = 0 .. nf) leaves them unwritten,
 * the caller solves those systems with the arrays of an earlier field (ffm_ldu_bind_coeffs_native_d with offDiagUnchanged). */
int ffm_fvm_scalar_transport_multi_w(ffm_mesh *m, int nf, const double *w_f, double rDeltaT, const double *rho, const double *rho0,
                                     const double *phi_f, const double *phi_b, const double *gamma_f, const double *gamma_b,
                                     const double *const *vf0, const double *const *f, const double *const *ref,
                                     const double *const *refGrad, const double *const *su, const double *const *su2,
                                     const double *const *sp, const double *const *expl3, double *const *diag,
                                     double *const *upper, double *const *lower, double *const *source);
/* momentum source of solver/UEqn.H:5 with `div(phi,U) Gauss LUST grad(U)`: source_c = rDeltaT*rho0*U0_c*V
 * - V*fvc::surfaceIntegrate(phi*LUST::correction(U_c)) for the three components from their gradients                  */
int ffm_fvm_lust_source3(ffm_mesh *m, double rDeltaT, const double *phi_f, const double *rho0, const double *const *U0,
                         const double *const *gx, const double *const *gy, const double *const *gz, double *const *source);

/* ------------------------------------------------------- fvDOM wall condition (N1) */
/* greyDiffusiveRadiationMixedFvPatchScalarField::updateCoeffs (reference packages/thermophysicalModels/radiation/derivedFvPatchFields/
 * greyDiffusiveRadiation/greyDiffusiveRadiationMixedFvPatchScalarField.C:150-230) of ray `ray` on every boundary face: with n the face
 * normal, nAve = n & dAve and Iw the ray's stored patch values,
 *   qr[ray] = Iw nAve;  Ir = sum over the rays of their qin (this ray's own, reset by radiativeIntensityRay::correct, counts 0);
 *   faces the ray leaves the wall through ((-n & d) > 0): valueFraction 1, refValue (Ir (1 - e) + e sigma T_b^4)/pi, qem[ray] = refValue nAve, qin[ray] = 0
 *   the others: valueFraction 0 (zeroGradient), qem[ray] = 0, qin[ray] = Iw nAve.
 * qin_all / qem_all / qr_all: device arrays [nRay][nBoundary] the rays share; emissivity_b NULL = 1 everywhere.  The iteration of
 * fvDOM::calculate (fvDOM/fvDOM.C:547-584) is host code above this (include/fireFoamHandles.H: fvDOM::correct).               */
int ffm_fvdom_wall_coeffs_d(ffm_mesh *mesh, int nRay, int ray, const double *d3, const double *dAve3, double sigma, const double *Iw_b,
                            const double *emissivity_b, const double *T_b, double *qin_all, double *qem_all, double *qr_all,
                            double *valueFraction_b, double *refValue_b);
/* fvDOM::updateG, boundary fluxes (fvDOM.C:740-750): out_b[k] = sum over the rays, in ray order, of all[ray][k] */
int ffm_fvdom_sum_rays_d(ffm_mesh *mesh, int nRay, const double *all, double *out_b);

/* ------------------------------------------------------- synthetic plume case */
/* Host-side driver (C++ over the entry points above) of one fireFoam time step on
 * the synthetic buoyant-plume box of SURVEY 8(d): rhoEqn, UEqn, YEEqn, 2 x pEqn in
 * the order of solver/fireFoam.C:97-119.  bench.py's workload.                */
typedef struct ffm_plume ffm_plume;
int ffm_plume_create(ffm_ctx *ctx, int nx, int ny, int nz, double h, double deltaT, ffm_plume **out);
/* one rank's block [lo,hi) of the global box; nbrRank[6] = rank across the -x,+x,-y,+y,-z,+z side or -1
 * (physical boundary).  Needs ffm_comm_init / ffm_comm_init_host when any neighbour exists.            */
int ffm_plume_create_block(ffm_ctx *ctx, int gx, int gy, int gz, const int *lo, const int *hi,
                           const int *nbrRank, double h, double deltaT, ffm_plume **out);
int ffm_plume_destroy(ffm_plume *p);
int ffm_plume_step(ffm_plume *p);
/* tests: run every linear solve to 1e-13 with relTol 0 instead of the fvSolution controls */
int ffm_plume_set_tight(ffm_plume *p, int on);
/* linear solvers of the transport equations (U, Yi, h): 0 = PBiCGStab + DILU (default, the kernels BASELINE names),
 * 1 = smoothSolver + symGaussSeidel with maxIter 10, the selection of cases/steckler/system/fvSolution:49-62 */
int ffm_plume_set_solvers(ffm_plume *plume, int stecklerSelection);
/* SURVEY 8(f) N1 stand-in: the reference's radiation->correct() (solver/YEEqn.H:80; fvDOM with nPhi 2, nTheta 4, solverFreq 100,
 * cases/steckler/constant/radiationProperties:32-40) as 4*nPhi*nTheta upwind ray-transport solves every `solverFreq` steps
 * (0 = off, the default).  dAve[3*nRay], omega[nRay]: the rays' mean directions and solid angles as fvDOM.C:55-90 builds
 * them, or both NULL to have them built from nPhi, nTheta.  Fields "G" and "I<n>" become readable with ffm_plume_get_field. */
int ffm_plume_set_radiation(ffm_plume *plume, int solverFreq, int nPhi, int nTheta, const double *dAve, const double *omega);
/* The reference's absorption / emission model for those rays and their coupling into the enthalpy equation:
 * constant absorption coefficient `absorption` [1/m] (constRadFractionEmission has aCont = 0: cases/steckler/constant/
 * radiationProperties:42; constantAbsorptionEmission: a) and the emission E = RadFraction*Qdot of lib/thermophysicalModels/
 * radiation/submodels/absorptionEmissionModel/constRadFractionEmission/constRadFractionEmission.C:ECont with radScaling:
 * RadFraction = max(min(Ehrr1, Ehrr2), (mlr1*Ehrr1 + mlr2*Ehrr2)/max(SMALL, mlr1 + mlr2)), mlr = -gSum(phi) of the burner patch
 * (cases/steckler: Ehrr1 0.5, Ehrr2 0.22).  Ray source omega/pi*(a sigma T^4 + E/4) (radiativeIntensityRay.C:286-300);
 * EEqn gets radiation->Sh(thermo, he) = Ru - fvm::Sp(4 Rp T^3/Cpv, he) - Rp T^3 (T - 4 he/Cpv), Rp = 4 a sigma, Ru = a G - E
 * (radiationModel.C:229-244; solver/YEEqn.H:101).  Without this call the rays keep the round-1 stand-in (a = 0.1, no E, no Sh). */
int ffm_plume_set_radiation_model(ffm_plume *plume, double absorption, double Ehrr1, double Ehrr2);
/* tests: a start state and boundary values other than the quiescent ambient / pure-fuel inflow -- Y[5] and h as cell fields in
 * natural blockMesh order, Yamb / Yin the inletOutlet and inlet values of the species, hAmb the inletOutlet value of h; redoes
 * the hydrostatic initialisation (solver/phrghEqn.H).  Single block, before the first step.                                   */
int ffm_plume_set_initial_state(ffm_plume *plume, const double *const *Y, const double *h, const double *Yamb, const double *Yin, double hAmb);
/* tests: the next step convects the species and h with the face weights w[F] (natural face order) instead of evaluating the
 * multivariateSelection limiter of solver/YEEqn.H:1-10 -- separates the limiter's conditioning from everything else in a step */
int ffm_plume_override_mv_weights(ffm_plume *plume, const double *w);
int ffm_plume_ncells(const ffm_plume *p);
int ffm_plume_nfaces(const ffm_plume *p);
int ffm_plume_get_field(ffm_plume *p, const char *name, double *out);
int ffm_plume_nsolves(const ffm_plume *p);
int ffm_plume_get_solve(const ffm_plume *p, int i, char *name16, ffm_perf *perf);
/* the case's raw state for a driver that runs the same case through another path (bench.py: the reference's equation files over the Foam
 * layer): cell fields [N] in the library's cell order, face fields [F] in the renumbered LDU face order, boundary arrays [B] in (patch, face)
 * order; returns the number of values or a negative FFM_ERR_*.  Names: rho p p_rgh h K dpdt gh Ux Uy Uz <specie> | phi ghf | phib ph_rgh_b kind
 * fStaticU<c> refU<c> fStaticS fStaticH refH refY<i> ghfB */
long ffm_plume_get_raw(ffm_plume *p, const char *name, double *out, long cap);
ffm_ldu *ffm_plume_ldu(ffm_plume *p);
ffm_mesh *ffm_plume_mesh(ffm_plume *p);          /* the case's device mesh (tests: operators on the tile-numbered mesh) */

/* ------------------------------------------------------- pyrolysis region (N3) */
/* reactingOneDim::evolveRegion (packages/regionModels/pyrolysisModels/reactingOneDim/reactingOneDim.C:686-721) for a panel of
 * nCol independent columns of nLay <= 16 layers (the region mesh extruded from a wall patch: cases/wallFireSpread2D/system/
 * extrudeToRegionMeshDict:17-39; cases/pyrolysis1D: one column, 8 layers): Arrhenius solid reaction, continuity, species and
 * the tridiagonal enthalpy equation of every column in one kernel, then solidThermo.correct().  Defaults: the solids and the
 * reaction of cases/pyrolysis1D/constant/panelRegion.  qSurf_d[nCol] (device): heat flux into the exposed face of every column
 * [W/m2] -- what the coupled wall conditions of lib/fvPatchFieldsPyrolysis hand over; the back face is adiabatic or held at Tback.
 * Out (device, [nCol]): the exposed cell's temperature and the pyrolysate mass flux phiGas [kg/s] of every column.            */
typedef struct ffm_pyro ffm_pyro;
int ffm_pyro_create(ffm_ctx *ctx, int nCol, int nLay, double thickness, double faceArea, double T0, double Yvirgin0, ffm_pyro **out);
int ffm_pyro_set_solids(ffm_pyro *p, const double *virgin /* rho Cp kappa Hf */, const double *charred);
int ffm_pyro_set_reaction(ffm_pyro *p, double A, double Ta, double Tcrit, double n);
int ffm_pyro_step(ffm_pyro *p, double deltaT, const double *qSurf_d, int backFixed, double Tback);
/* name: rho, Yw, T, h, alpha -> host [nCol][nLay]; Tsurf, phiGas, qSurf, Twall -> host [nCol] */
int ffm_pyro_get(ffm_pyro *p, const char *name, double *out);
const double *ffm_pyro_surface_T_d(const ffm_pyro *p);
const double *ffm_pyro_phiGas_d(const ffm_pyro *p);
/* The mapped patch conditions between the gas region's wall patch and the panel (lib/fvPatchFieldsPyrolysis), column i <->
 * gas boundary face map_d[i] (null: i).  Replaces turbulentTemperatureRadiationQinCoupledMixedFvPatchScalarField::updateCoeffs
 * (:176-292; solid side: qSurf = -(kappaDelta_gas (T_s,cell - T_g,cell) - a qin + e sigma T_w^4), wall value from refGrad;
 * gas side: refValue = T_s,cell, valueFraction 1 -> refT_d) and flowRateInletVelocityPyrolysisCoupledFvPatchVectorField::
 * updateCoeffs (:127-248; U_b = n (-phiGas hocPyr/qFuel/magSf)/rho_b -> Ux_d, Uy_d, Uz_d).  Gas-side arrays are indexed by
 * boundary face; only the mapped entries of the outputs are written.  ffm_pyro_qSurf_d is what ffm_pyro_step then takes. */
int ffm_pyro_couple_d(ffm_pyro *p, const int *map_d, const double *TgasCell_d, const double *kappaDelta_d, const double *qin_d,
                      double emissivity, double absorptivity, const double *rho_b_d, const double *magSf_d,
                      const double *nfx_d, const double *nfy_d, const double *nfz_d, double hocSolid, double qFuel,
                      double *refT_d, double *Ux_d, double *Uy_d, double *Uz_d);
const double *ffm_pyro_qSurf_d(const ffm_pyro *p);
/* Selections of the region's dictionaries.
 * ffm_pyro_set_model: reactingOneDim21 != 0 selects lib/regionModels/pyrolysisModels/reactingOneDim21/reactingOneDim21.C (solveEnergy
 *   :319-368: fvm::ddt(rho,h) - fvm::laplacian(alpha,h) + fvc::laplacian(alpha,h) - fvc::laplacian(kappa,T) == chemistryQdot +
 *   RRs(0) T Cp0 + RRs(1) T Cp1; evolveRegion :777-815), the pyrolysisModel of cases/wallFireSpread2D/constant/pyrolysisZones:23,
 *   instead of reactingOneDim (== chemistryQdot - fvm::Sp(RRg, h)); harmonicAlpha / harmonicKappa: `Gauss harmonic` instead of
 *   `Gauss linear` for laplacian(thermo:alpha,h) / laplacian(kappa,T) (system/panelRegion/fvSchemes of the two cases).
 * ffm_pyro_set_back: the back face of T -- 0 zeroGradient, 1 fixedValue Tinf, 2 constHTemperature (lib/fvPatchFields/
 *   constHTemperatureFvPatchScalarField/constHTemperatureFvPatchScalarField.C:156-180; cases/wallFireSpread2D/0/panelRegion/T:25-31).
 * ffm_pyro_set_surface_radiation: greyMeanSolidAbsorptionEmission (cases/wallFireSpread2D/constant/panelRegion/radiationProperties:
 *   19-31): absorptivity / emissivity of the exposed face = the solids' values weighted with their volume fractions in the exposed
 *   layer; used by ffm_pyro_evolve_d instead of its constants and handed to the gas side by ffm_pyro_gas_side_d.                      */
int ffm_pyro_set_model(ffm_pyro *p, int reactingOneDim21, int harmonicAlpha, int harmonicKappa);
int ffm_pyro_set_back(ffm_pyro *p, int mode, double h, double Tinf);
int ffm_pyro_set_surface_radiation(ffm_pyro *p, double absorptivityVirgin, double emissivityVirgin, double absorptivityChar, double emissivityChar);
/* pyrolysisModel::evolve() of one region with its exposed face coupled to the gas region (solver/fireFoam.C:90-93): evolveRegion with
 * the solid branch of turbulentTemperatureRadiationQinCoupledMixedFvPatchScalarField::updateCoeffs (:176-296) evaluated INSIDE the
 * step, where OpenFOAM evaluates it (construction of hEqn: old cell temperature, stored wall value, surface properties and
 * kappa(*this) of the composition after solveSpeciesMass); the wall value and qSurf of the step are stored.  Gas-side arrays
 * (cell temperature next to the wall, kappaEff*deltaCoeffs, incident radiation qin or NULL) are indexed by boundary face,
 * column i <-> face map_d[i] (NULL: i).  emissivity / absorptivity: constants used unless ffm_pyro_set_surface_radiation was called. */
int ffm_pyro_evolve_d(ffm_pyro *p, double deltaT, const int *map_d, const double *TgasCell_d, const double *kappaDelta_d, const double *qin_d,
                      double emissivity, double absorptivity);
/* what the gas region's wall patch reads from the panel after its evolve: refT_d (fluid branch of the coupled condition, :297-303),
 * U_b (flowRateInletVelocityPyrolysisCoupledFvPatchVectorField::updateCoeffs :127-248) and, if emissivity_d is given and the surface
 * radiation model is set, the wall emissivity of greyDiffusiveRadiation with `emissivityMode solidRadiation`
 * (packages/thermophysicalModels/radiation/derivedFvPatchFields/radiationCoupledBase/radiationCoupledBase.C:150-182).  Only the
 * mapped entries are written.                                                                                                      */
int ffm_pyro_gas_side_d(ffm_pyro *p, const int *map_d, const double *rho_b_d, const double *magSf_d, const double *nfx_d, const double *nfy_d,
                        const double *nfz_d, double hocSolid, double qFuel, double *refT_d, double *Ux_d, double *Uy_d, double *Uz_d,
                        double *emissivity_d);
/* reactingOneDim21::solidRegionDiffNo (reactingOneDim21.C:697-714; solver/solidRegionDiffusionNo.H): max over the region's internal faces of
 * deltaCoeffs^2 interpolate(kappa())/interpolate(Cp() rho) * deltaT -- the diffusion number solver/setMultiRegionDeltaT.H limits with maxDi */
int ffm_pyro_diff_no(ffm_pyro *p, double deltaT, double *out);
int ffm_pyro_destroy(ffm_pyro *p);

/* ------------------------------------------------------- thermo, combustion, LES (N2) */
/* The per-cell physics the reference's time step calls between its equations, one streaming kernel each.
 * ffm_thermo_*: hePsiThermo<reactingMixture<sutherland<janaf<perfectGas<specie>>>>, sensibleEnthalpy> (cases/steckler/constant/
 *   thermophysicalProperties:18-27, thermo.compressibleGas): replaces thermo.correct() (solver/YEEqn.H:114) = hePsiThermo::
 *   calculate() -- cell (or patch-face) mixture by mass fractions, T from he by thermo::T's Newton iteration started at the old
 *   T, psi = 1/(R T), Sutherland mu, alpha = kappa(modified Eucken)/Cp -- and he = Hs(T) where a patch fixes T.  Species table as
 *   in the thermo file (molar coefficients; highCpCoeffs / lowCpCoeffs [nSpecies][7]); Y_d: nSpecies device pointers.
 * ffm_edc_correct_d: the reference's eddyDissipationModel::correct (lib/thermophysicalModels/combustionModels/
 *   eddyDissipationModel/eddyDissipationModel.C:93-149) with kEqn::epsilon = Ce k sqrt(k)/delta: wFuel and Qdot = qFuel wFuel.
 * ffm_les_keqn_nut_d: kEqn::correctNut (nut = Ck sqrt(k) delta) and alphat = rho nut/Prt (alphat_d null: nut only). */
typedef struct ffm_thermo ffm_thermo;
int ffm_thermo_create(ffm_ctx *ctx, int nSpecies, const double *W, const double *Tlow, const double *Thigh, const double *Tcommon,
                      const double *highCpCoeffs, const double *lowCpCoeffs, const double *As, const double *Ts, double RR,
                      ffm_thermo **out);
int ffm_thermo_correct_d(ffm_thermo *th, long n, const double *const *Y_d, const double *he_d, const double *p_d, double *T_d,
                         double *psi_d, double *mu_d, double *alpha_d);
int ffm_thermo_he_d(ffm_thermo *th, long n, const double *const *Y_d, const double *T_d, double *he_d);
/* psi, mu, alpha (each nullable) of the mixture at a given temperature, no iteration (patch faces whose T is fixed) */
int ffm_thermo_properties_d(ffm_thermo *th, long n, const double *const *Y_d, const double *T_d, double *psi_d, double *mu_d,
                            double *alpha_d);
/* Cp of the mixture at T (heThermo::Cp(p, T, patchi): kappaEff of compressible::thermalBaffle1D -- cases/steckler/0/T:51-82 --
 * and the gradient term of mixedEnergy::updateCoeffs) */
int ffm_thermo_Cp_d(ffm_thermo *th, long n, const double *const *Y_d, const double *T_d, double *Cp_d);
int ffm_thermo_destroy(ffm_thermo *th);
int ffm_edc_correct_d(ffm_ctx *ctx, long n, const double *rho_d, const double *k_d, const double *delta_d, const double *alpha_d,
                      const double *Yfuel_d, const double *YO2_d, double s, double deltaT, double Ce, double C_EDC, double C_Diff,
                      double C_Stiff, double qFuel, double *wFuel_d, double *Qdot_d);
int ffm_les_keqn_nut_d(ffm_ctx *ctx, long n, double Ck, double Prt, const double *k_d, const double *delta_d, const double *rho_d,
                       double *nut_d, double *alphat_d);

/* turbulence->divDevRhoReff(U) (solver/UEqn.H:12), explicit part: out[j] = component j of fvc::div(gamma*dev2(T(fvc::grad(U)))),
 * Gauss linear, from the nine cell gradients g[3*i + j] = d_i U_j (ffm_fvc_grad_multi of the three components), gamma = rho*nuEff
 * on cells / patch faces, U and its patch values (the boundary gradient is gaussGrad's corrected one).  The implicit part is
 * ffm_fvm_transport with gamma_f.  ffm_les_keqn_G: G = nut*(gradU && dev(twoSymm(gradU))) of kEqn::correct. */
int ffm_fvc_div_dev2T_gradU(ffm_mesh *m, const double *const *g, const double *gamma, const double *gamma_b,
                            const double *const *U, const double *const *U_b, double *const *out);
int ffm_les_keqn_G(ffm_mesh *m, const double *const *g, const double *nut, double *G);

/* ------------------------------------------------------------------------ GAMG */
/* pairGAMGAgglomeration::forward_ is a static upstream: the cell-visiting direction of the next pair agglomeration, toggled by every
 * agglomeration of the run (a second GAMG mesh or region continues where the first ended).  Here it lives in the context (= the run):
 * ffm_gamg_create reads and updates it; these two set / read it (a fresh context starts forward, as a fresh run does).               */
int ffm_ctx_set_gamg_forward(ffm_ctx *ctx, int forward);
int ffm_ctx_gamg_forward(const ffm_ctx *ctx);
/* lduMatrix::solver::New(... solver GAMG ...) as the reference's dictionaries select it: agglomerator faceAreaPair,
 * mergeLevels 1, nCellsInCoarsestLevel 10, cacheAgglomeration true, smoother GaussSeidel for p_rgh / ph_rgh
 * (cases/wallFireSpread2D/system/fvSolution:36-60, cases/pyrolysis1D/system/fvSolution) and DILU for Ii
 * (cases/steckler/system/fvSolution:63-73).  Replaces OpenFOAM-dev's GAMGSolver::solve + pairGAMGAgglomeration (the
 * algorithm is restated with its sources in oracle/gamg.py).  Decomposed meshes: `finest` is a ghost-cell matrix
 * (ffm_ldu_create_ext + ffm_ldu_set_ghost_exchange; nCells = owned + ghost, the cut faces in lowerAddr / upperAddr); every rank
 * agglomerates its own cells, the processor interfaces are agglomerated with them (coarse ghost cells, coarse cut faces, a ghost
 * exchange per level), continueAgglomerating / normFactor / residuals / scale factors are global -- OpenFOAM's behaviour without a
 * processorAgglomerator (oracle/gamg_multi.py; tests/test_partition_gpu.py::test_gamg_on_a_decomposed_mesh).  A mesh below
 * nCellsInCoarsestLevel gets no coarse level: the solve then is the coarsest-level solver on the fine matrix.
 * ffm_gamg_create: the agglomeration of the mesh behind `finest` (created from the same lowerAddr / upperAddr), built once
 *   (cacheAgglomeration); faceWeights [nFaces] host, e.g. from ffm_gamg_face_area_pair_weights(Sf [nFaces][3]).
 * ffm_gamg_set_matrix_d: GAMGSolver::agglomerateMatrix for every level, from device coefficient arrays in the caller's
 *   cell / face order (lower_d null: symmetric); also sets the coefficients of `finest`.  The arrays must stay alive until
 *   the next call.
 * ffm_gamg_solve_d: GAMGSolver::solve; smoother FFM_GS (GaussSeidel), FFM_SYMGS, FFM_DIC or FFM_DILU; nPreSweeps 0,
 *   nPostSweeps 2 (+1 per level, at most 4), nFinestSweeps 2 (ffm_gamg_set_sweeps), scaleCorrection on symmetric matrices,
 *   coarsest level by PCG+DIC / PBiCGStab+DILU to the same tolerance and relTol.  psi_d / source_d: device, caller order. */
typedef struct ffm_gamg ffm_gamg;
int ffm_gamg_face_area_pair_weights(int nFaces, const double *Sf, double *weights);
int ffm_gamg_create(ffm_ctx *ctx, ffm_ldu *finest, int nCells, int nFaces, const int *lowerAddr, const int *upperAddr,
                    const double *faceWeights, int nCellsInCoarsestLevel, int mergeLevels, ffm_gamg **out);
int ffm_gamg_set_sweeps(ffm_gamg *g, int nPreSweeps, int nPostSweeps, int nFinestSweeps);
int ffm_gamg_set_matrix_d(ffm_gamg *g, const double *diag_d, const double *upper_d, const double *lower_d);
/* the same from the library's native coefficient layout (ffm_ldu_set_coeffs_native_d; the fvMatrix of include/ffmFoam.H) */
int ffm_gamg_set_matrix_native_d(ffm_gamg *g, const double *diag_d, const double *upperNative_d, const double *lowerNative_d);
int ffm_gamg_solve_d(ffm_gamg *g, int smoother, double tolerance, double relTol, int minIter, int maxIter,
                     double *psi_d, const double *source_d, ffm_perf *out);
int ffm_gamg_nlevels(const ffm_gamg *g);                                     /* coarse levels                              */
int ffm_gamg_level_size(const ffm_gamg *g, int level, int *nCells, int *nFaces);          /* level 0: the caller's matrix */
int ffm_gamg_get_level_addressing(const ffm_gamg *g, int level, int *lowerAddr, int *upperAddr);
int ffm_gamg_get_level_coeffs(ffm_gamg *g, int level, double *diag, double *upper, double *lower);   /* host, level order */
int ffm_gamg_coarsest_solves(const ffm_gamg *g, int maxN, ffm_perf *out);    /* of the last solve; returns their number    */
int ffm_gamg_destroy(ffm_gamg *g);

int ffm_device_synchronize(void);     /* hipDeviceSynchronize (debugging aid of the Foam layer: FFM_SYNC_RANGE) */

/* ---------------------------------------------------------------- reductions */
/* gSum / gMin / gMax / gSumProd / gSumMag over a device field (solver/YEEqn.H:
 * 73-78,117-118; solver/phrghEqn.H:54-55).  All-reduced when a communicator is
 * attached.                                                                  */
int ffm_reduce_sum(ffm_ctx *ctx, const double *x_d, long n, double *out);
int ffm_reduce_min(ffm_ctx *ctx, const double *x_d, long n, double *out);
int ffm_reduce_max(ffm_ctx *ctx, const double *x_d, long n, double *out);
int ffm_reduce_dot(ffm_ctx *ctx, const double *x_d, const double *y_d, long n, double *out);
int ffm_reduce_summag(ffm_ctx *ctx, const double *x_d, long n, double *out);

/* ---------------------------------------------------------------------- comm */
/* Pstream replacement: one process per GPU, RCCL over xGMI.
 * uniqueId = the 128-byte ncclUniqueId made by rank 0 (ffm_comm_unique_id) and
 * distributed by the host launcher (torch.distributed / MPI / a file).        */
int ffm_comm_unique_id(void *uniqueId128);
int ffm_comm_init(ffm_ctx *ctx, int rank, int nRanks, const void *uniqueId128);
/* Alternative transport through the host launcher (MPI, gloo, ...): lets several
 * ranks share one GPU (single-GPU testing of the decomposed path).
 * allreduce: in-place over n doubles, op 0 = sum, 1 = min, 2 = max.
 * exchange: patch p sends sendBuf[offset[p] .. +size[p]) to rank nbrRank[p] and
 * receives the same count from it into recvBuf at the same offset.           */
typedef void (*ffm_host_allreduce_fn)(void *user, double *vals, int n, int op);
typedef void (*ffm_host_exchange_fn)(void *user, int nPatches, const int *size,
                                     const int *nbrRank, const int *offset,
                                     const double *sendBuf, double *recvBuf);
int ffm_comm_init_host(ffm_ctx *ctx, int rank, int nRanks, void *user,
                       ffm_host_allreduce_fn allreduce,
                       ffm_host_exchange_fn exchange);
/* variable-count form used by the ghost-cell halo: neighbour q sends
 * sendBuf[sendOff[q]..sendOff[q+1]) to nbrRank[q] and receives recvBuf[recvOff[q]..recvOff[q+1]) from it */
typedef void (*ffm_host_exchange2_fn)(void *user, int nNbr, const int *nbrRank, const int *sendOff,
                                      const int *recvOff, const double *sendBuf, double *recvBuf);
int ffm_comm_set_host_exchange2(ffm_ctx *ctx, ffm_host_exchange2_fn fn);
int ffm_comm_rank(const ffm_ctx *ctx);
int ffm_comm_size(const ffm_ctx *ctx);

/* Diagnostics of the tiled sweeps (ffm_tile.hip): the first call switches tracing on and returns the number of groups; later
 * calls copy out, for the last tiled launch, 4 words per group {start, first entry ready, end (100 MHz ticks), mailbox
 * re-loads}.  Not part of the reference interface. */
int ffm_debug_tile_trace(ffm_ldu *A, unsigned long long *out, int nWords);
/* tests: preset the group ticket counter of the tiled sweeps (every sweep launch zeroes it again on the stream, so a preset
 * close to 2^32 must not change any result)                                                                             */
int ffm_debug_set_sweep_ticket(ffm_ldu *A, unsigned int value);

/* ------------------------------------------------------- decomposition (host) */
/* decomposePar's job for this path (reference: scotch, cases/steckler/system/decomposeParDict:18-20; `simple`,
 * cases/wallFireSpread2D/system/decomposeParDict:18-27; BASELINE names METIS -- none of the three libraries is in this image).
 * part[nCells] = sub-domain of every cell:
 *   ffm_partition_rcb    recursive coordinate bisection of the cell centres C[3][nCells], any number of parts
 *   ffm_partition_graph  greedy graph growing on the LDU graph (no geometry needed)                                        */
int ffm_partition_rcb(int nCells, const double *C, int nParts, int *part);
int ffm_partition_graph(int nCells, int nFaces, const int *lowerAddr, const int *upperAddr, int nParts, int *part);
/* The sub-domain of `rank` in the ghost-cell form of ffm_ldu_create_ext: owned cells in their global relative order, then the
 * ghost cells grouped by neighbour rank (ascending global label inside a group); faces in upper-triangular order, a cut face
 * owned by its owned cell and flagged `flip` where the global owner is the ghost (its local upper coefficient is the global
 * lower one).  ffm_subdomain_exchange gives the arguments of ffm_ldu_set_ghost_exchange and the pair tags of
 * ffm_ldu_set_exchange_tags (kind 1); ffm_subdomain_cut_faces lists the cut faces per neighbour in ascending global face
 * label -- the processor-patch form (faceCells per patch, same order on both sides) for ffm_ldu_set_interfaces.           */
typedef struct ffm_subdomain ffm_subdomain;
int ffm_subdomain_create(int nCells, int nFaces, const int *lowerAddr, const int *upperAddr, const int *part, int nParts,
                         int rank, ffm_subdomain **out);
int ffm_subdomain_destroy(ffm_subdomain *s);
int ffm_subdomain_sizes(const ffm_subdomain *s, int *nOwned, int *nGhost, int *nFaces, int *nNbr, int *nSend, int *nCut);
int ffm_subdomain_cells(const ffm_subdomain *s, int *globalCell /* [nOwned+nGhost] */);
int ffm_subdomain_faces(const ffm_subdomain *s, int *lowerAddr, int *upperAddr, int *globalFace, int *flip);
int ffm_subdomain_exchange(const ffm_subdomain *s, int *nbrRank, int *sendCount, int *sendCells, int *recvCount, int *tags);
int ffm_subdomain_cut_faces(const ffm_subdomain *s, int *cutStart /* [nNbr+1] */, int *localCell, int *globalFace, int *flip);

/* ------------------------------------------------------- polyMesh (host)    */
/* SURVEY 8(f) N4, first part: OpenFOAM's on-disk mesh (ascii constant/polyMesh/{points,faces,owner,neighbour,boundary}, as
 * blockMesh / topoSet / createBaffles leave it: reference cases/steckler/mesh.sh:8-21) and the finite-volume geometry
 * derived from it with OpenFOAM's algorithms (face fans, cell pyramids, weights, nonOrthDeltaCoeffs, correction vectors):
 * the inputs of ffm_ldu_create / ffm_mesh_create / ffm_mesh_set_face_centres / ffm_mesh_set_nonorth_correction.  Host only. */
typedef struct ffm_polymesh ffm_polymesh;
int ffm_polymesh_read(const char *polyMeshDir, ffm_polymesh **out);
int ffm_polymesh_destroy(ffm_polymesh *pm);
int ffm_polymesh_sizes(const ffm_polymesh *pm, int *nPoints, int *nCells, int *nFaces, int *nInternalFaces, int *nPatches);
int ffm_polymesh_addressing(const ffm_polymesh *pm, int *lowerAddr, int *upperAddr);
/* cell arrays [N] / [3][N]; internal-face arrays [F] / [3][F]; NULL = not wanted */
int ffm_polymesh_geometry(const ffm_polymesh *pm, double *V, double *C, double *Sf, double *Cf, double *magSf, double *weights,
                          double *nonOrthDeltaCoeffs, double *nonOrthCorrectionVectors);
int ffm_polymesh_patch(const ffm_polymesh *pm, int i, char *name64, char *type32, int *startFace, int *nFaces);
/* processorN/constant/polyMesh of a case decomposed by decomposePar: 1 if patch i is `type processor` (then *myProcNo and
 * *neighbProcNo hold its entries), 0 if not.  Its faceCells (ffm_polymesh_patch_geometry) and the neighbour rank are the
 * faceCells / neighbRank of ffm_ldu_set_interfaces; both sides list the faces of a processor patch in the same order. */
int ffm_polymesh_patch_processor(const ffm_polymesh *pm, int i, int *myProcNo, int *neighbProcNo);
int ffm_polymesh_patch_geometry(const ffm_polymesh *pm, int i, int *faceCells, double *Sf, double *Cf, double *deltaCoeffs);

#ifdef __cplusplus
}
#endif
#endif
