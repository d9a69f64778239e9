// ffm_plume.hip -- the synthetic buoyant-plume case of SURVEY 8(d): host-side driver of one
// fireFoam time step (solver/fireFoam.C:76-121 with PIMPLE 1/2/0, cases/steckler/system/
// fvSolution:84-89) written against the C ABI of include/ffm.h, in the order of the reference's
// equation snippets:
//   rhoEqn  solver/rhoEqn.H:33-43      UEqn  solver/UEqn.H:3-33
//   YEEqn   solver/YEEqn.H:37-118      pEqn  solver/pEqn.H:1-60 (x2)   start-up solver/phrghEqn.H:25-56
// This is bench.py's workload and the subject of tests/test_plume_gpu.py (oracle: oracle/plume.py,
// same sequence in numpy).  The physics plug-ins that the reference takes from other libraries are
// replaced by the stand-ins listed in oracle/plume.py (perfect gas / constant Cp, constant mu, Pr,
// EDC-shaped single-step source, zero-gradient thermo boundary values); their kernels are the
// k_standin_* below and are not part of the hot path being reproduced.
#include "ffm_internal.hpp"
#include "ffm_device.hpp"
#include <algorithm>
#include <cmath>
#include <string>

struct ffm_mesh;
extern "C" {
int ffm_ldu_bind_coeffs_native_d(ffm_ldu *, const double *, const double *, const double *, int);
int ffm_mesh_create(ffm_ldu *, const double *, const double *, const double *, const double *, const double *, const double *, int,
                    const int *, const int *const *, const double *const *, const double *const *, ffm_mesh **);
int ffm_mesh_destroy(ffm_mesh *);
int ffm_mesh_set_face_centres(ffm_mesh *, const double *);
int ffm_fv_lust_correction(ffm_mesh *, const double *, const double *, const double *, const double *, double *);
int ffm_fvc_interpolate(ffm_mesh *, const double *, const double *, double *);
int ffm_fvc_snGrad(ffm_mesh *, const double *, double *);
int ffm_fvc_snGrad_b(ffm_mesh *, const double *, const double *, double *);
int ffm_fvc_flux(ffm_mesh *, const double *, const double *, const double *, double *);
int ffm_fvc_surface_integrate(ffm_mesh *, const double *, const double *, double *);
int ffm_fvc_grad(ffm_mesh *, const double *, const double *, double *, double *, double *);
int ffm_fvc_reconstruct(ffm_mesh *, const double *, const double *, double *, double *, double *);
int ffm_fv_limited_weights(ffm_mesh *, int, double, double, double, const double *, const double *, const double *, const double *,
                           const double *, double *);
int ffm_fvm_transport(ffm_mesh *, double, const double *, const double *, const double *, const double *, int, double *, double *, double *);
int ffm_fvm_boundary_coeffs(ffm_mesh *, const double *, const double *, int, const double *, const double *, const double *, double *, double *);
int ffm_bc_values(ffm_mesh *, const double *, const double *, const double *, const double *, double *);
int ffm_fvm_add_boundary(ffm_mesh *, const double *, const double *, const double *, const double *, const double *, double *, double *);
int ffm_fvm_A(ffm_mesh *, int, const double *, const double *, const double *, const double *, double *);
int ffm_fvm_H(ffm_mesh *, int, int, const double *, const double *, const double *, const double *, const double *, const double *,
              const double *, const double *, double *);
int ffm_fvm_flux(ffm_mesh *, const double *, const double *, const double *, const double *, const double *, double *, double *);
int ffm_fvc_grad_multi(ffm_mesh *, int, const double *const *, const double *const *, double *const *, double *const *, double *const *);
int ffm_fvm_scalar_transport_multi(ffm_mesh *, int, int, double, double, double, double, const double *, const double *, const double *,
                                   const double *, const double *, const double *, const double *const *, const double *const *,
                                   const double *const *, const double *const *, const double *const *, const double *const *,
                                   const double *const *, const double *const *, const double *const *, const double *const *,
                                   const double *const *, const double *const *,
                                   double *const *, double *const *, double *const *, double *const *);
int ffm_pc_phig(ffm_mesh *, const double *, const double *, const double *, double *);
int ffm_ue_buoyancy_flux(ffm_mesh *, const double *, const double *, const double *, double *);
int ffm_pc_phiHbyA(ffm_mesh *, const double *, const double *, const double *, const double *, const double *, const double *, const double *, double *);
int ffm_pc_flux(ffm_mesh *, const double *, const double *, const double *, const double *, const double *, const double *, double *, double *, double *);
int ffm_fv_limited_limiter(ffm_mesh *, int, double, double, double, const double *, const double *, const double *, const double *,
                           const double *, double *, int);
int ffm_fv_weights_from_limiter(ffm_mesh *, const double *, const double *, double *);
int ffm_fv_multivariate_weights(ffm_mesh *, int, const int *, double, double, double, const double *, const double *const *,
                                const double *const *, const double *const *, const double *const *, double *);
int ffm_fvm_scalar_transport_multi_w(ffm_mesh *, int, const double *, double, const double *, const double *, const double *, const double *,
                                     const double *, const double *, const double *const *, const double *const *, const double *const *,
                                     const double *const *, const double *const *, const double *const *, const double *const *,
                                     const double *const *, double *const *, double *const *, double *const *, double *const *);
int ffm_fvm_lust_source3(ffm_mesh *, double, const double *, const double *, const double *const *, const double *const *,
                         const double *const *, const double *const *, double *const *);
int ffm_fv_multivariate_weights_tiled(ffm_mesh *, int, const int *, double, double, double, const double *, const double *const *,
                                      const double *const *, double *);
}
const int *ffm_mesh_bcells(const ffm_mesh *m);
const double *ffm_mesh_geom(const ffm_mesh *m, int which);

namespace {
constexpr double RR = 8314.47, CP = 1005.0, TREF = 298.15, PREF = 101325.0, MU = 1.8e-5, PR = 0.7;
constexpr double S_O2 = 3.6282945, HC = 46357151.0, TAU = 0.05, T_IN = 600.0, U_IN = 0.5;
constexpr int NSP = 5, INERT = 4;
const double WMOL[NSP] = {31.9988, 18.0153, 44.0962, 44.01, 28.0134};
const double Y_AMB[NSP] = {0.23301, 0.0, 0.0, 0.0, 0.76699};
const double Y_IN[NSP] = {0.0, 0.0, 1.0, 0.0, 0.0};
const double NU[NSP] = {-S_O2, 4 * 18.0153 / 44.0962, -1.0, 3 * 44.01 / 44.0962, 0.0};
const char *SPN[NSP] = {"O2", "H2O", "C3H8", "CO2", "N2"};
enum { P_INLET = 0, P_FLOOR = 1, P_TOP = 2, P_SIDES = 3 };

template <class F> __global__ void k_for(long n, F f)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) f(i);
}
inline int sgrid(long n) { long g = (n + 255) / 256; return (int)std::max(1L, std::min(g, (long)RED_BLOCKS)); }
}  // namespace

struct SolveLog { char name[16]; ffm_perf perf; };

struct ffm_plume {
    ffm_ctx *ctx = nullptr; ffm_ldu *A = nullptr; ffm_mesh *mesh = nullptr;
    int nx = 0, ny = 0, nz = 0, N = 0, nOwn = 0, F = 0, nNat = 0, B = 0;   // N = owned + ghost cells
    double h = 0.05, dt = 1e-3, rdt = 1e3, time = 0.0;
    std::vector<int> newToOld;             // library cell order -> natural blockMesh cell id
    std::vector<int> faceNewToOld;         // library (caller-side) face order -> natural blockMesh face id
    bool mvOverride = false;               // tests: the next step convects the species and h with weights handed in (ffm_plume_override_mv_weights)
    double hAmb = 0.0;                     // inletOutlet reference value of h on the open patches
    std::vector<double *> pool;            // every device buffer, for destroy
    // fields
    double *Y[NSP], *Y0[NSP], *T, *hs, *hs0, *U[3], *U0[3], *p, *p0, *p_rgh, *p_rgh0, *psi, *psi0, *rho, *rho0, *K, *K0, *dpdt;
    double *phi, *phi0, *phib, *phib0, *gh, *ghf, *ph_rgh, *ph_rgh_b;
    // boundary-condition data [B]
    double *kind_d;                        // patch kind per boundary face (as double for simple kernels)
    double *fU[3], *refU[3], *fS, *refS, *fP, *refP, *gradP, *zeroB, *oneB;
    double *fStaticU[3], *fStaticS, *fStaticH, *fH, *refY[NSP], *refH;      // static templates (-1 = inletOutlet)
    // matrix + work
    double *diag, *upper, *lower, *src[3], *ic[3], *bc[3], *dWork, *sWork;
    double *UdW[3] = {nullptr, nullptr, nullptr}, *UsW[3] = {nullptr, nullptr, nullptr};   // diagonal + source of the three components (one lock-step solve)
    double *wN[12], *wF[6], *wB[8];
    double *ddtCorrF = nullptr; bool ddtCorrValid = false;      // coeff*rDeltaT*phiCorr of fvc::ddtCorr(rho, U, phi): old-time fields only, the same in both correctors of a step
    // fused assembly (ffm_fused.hip): gradients of up to 4 fields, the matrices of the 4 transported species, their patch values
    double *gM[4][3], *spD[4], *spU[4], *spL[4], *spS[4], *spB[4], *suM[4];
    // mvConvection of solver/YEEqn.H:1-10 (`Gauss multivariateSelection`): the weights of the ONE limiter all species and h are
    // convected with -- the minimum of the member schemes' limiters over the five species and h (FFM_PLUME_INDEPENDENT_LIMITERS=1:
    // one limiter per field, as round 1 had it)
    bool mvSelection = true;
    double *wMv = nullptr, *mvG[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    bool fused = true;                     // FFM_PLUME_UNFUSED: one kernel per operator (tests compare the two paths)
    // UEqn kept for pEqn (A, H)
    double *Udiag, *Uupper, *Ulower, *Usrc[3], *Uic[3], *Ubc[3];
    std::vector<SolveLog> log;
    bool tight = false;                    // tests: every solve to 1e-13 / relTol 0 (removes the stopping-rule noise)
    // fvDOM stand-in: off unless ffm_plume_set_radiation() was called
    int stepNo = 0, radFreq = 0;
    std::vector<double> rayD, rayOmega;    // dAve[3] and omega per ray
    // direction-ordered ray solves (single block): for the flip of axis a, radCm[a][c] = cell that takes c's place and
    // radFm[a][e] = native face that takes e's place (bit-complemented where owner and neighbour change roles); the permuted
    // system has the sparsity of A and is triangular for every ray whose minority-sign axis is a, so DILU solves it exactly
    std::vector<int> hL2, hU2, hOldToNew; std::vector<signed char> hFd2;
    int *radCm[3] = {nullptr, nullptr, nullptr}, *radFm[3] = {nullptr, nullptr, nullptr};
    double *radDB = nullptr, *radSB = nullptr, *radPsiB = nullptr, *radUB = nullptr, *radLB = nullptr;
    bool radOrdered = false;
    std::vector<double *> I; double *G = nullptr, *radJ = nullptr, *radW = nullptr, *radJb = nullptr, *radF = nullptr, *radRef = nullptr, *radSrc = nullptr;
    // the reference's absorption / emission model and radiation->Sh (ffm_plume_set_radiation_model): constant absorption
    // coefficient, emission E = RadFraction*Qdot with the radScaling of constRadFractionEmission::ECont
    bool radCoupled = false; double radA = 0.1, Ehrr1 = 0.0, Ehrr2 = 0.0;
    double *radE = nullptr, *radShSu = nullptr, *radShSp = nullptr; bool radHaveG = false;
    bool stecklerSolvers = false;          // transport equations with smoothSolver + symGaussSeidel, maxIter 10
                                           // (cases/steckler/system/fvSolution:49-62) instead of PBiCGStab + DILU
};

#define PL_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { ffm_set_error("%s:%d %s", __FILE__, __LINE__, hipGetErrorString(e_)); return FFM_ERR_HIP; } } while (0)

static double *dalloc(ffm_plume *P, size_t n)
{
    double *p = nullptr;
    if (hipMalloc((void **)&p, sizeof(double) * std::max<size_t>(n, 1)) != hipSuccess) return nullptr;
    if (ffm_dzero(P->ctx, p, sizeof(double) * std::max<size_t>(n, 1)) != FFM_OK) { hipFree(p); return nullptr; }
    P->pool.push_back(p);
    return p;
}
static double *dupload(ffm_plume *P, const std::vector<double> &v)
{
    double *p = dalloc(P, v.size());
    if (p && !v.empty() && ffm_h2d(P->ctx, p, v.data(), sizeof(double) * v.size()) != FFM_OK) return nullptr;
    return p;
}
template <class Fn> static void forN(ffm_plume *P, long n, Fn f)
{
    if (n > 0) hipLaunchKernelGGL(k_for<Fn>, dim3(sgrid(n)), dim3(256), 0, P->ctx->stream, n, f);
}
static double *dfill(ffm_plume *P, size_t n, double value)          // a field with one value everywhere, filled on the device
{
    double *p = dalloc(P, n);
    if (p && n) forN(P, (long)n, [=] __device__(long i) { p[i] = value; });
    return p;
}
static void dcopy(ffm_plume *P, double *d, const double *s, long n)
{ hipMemcpyAsync(d, s, sizeof(double) * n, hipMemcpyDeviceToDevice, P->ctx->stream); }

// refresh the ghost-cell entries of a cell field from the neighbour ranks (no-op on a single rank)
static int HX(ffm_plume *P, double *f) { return (P->N > P->nOwn) ? ffm_halo_refresh_d(P->A, f) : FFM_OK; }

// ---- stand-in physics (not part of the reproduced hot path) -----------------------------------
static void standin_thermo(ffm_plume *P)
{   // T = Tref + h/Cp ; psi = 1/(R T sum(Y_i/W_i))
    double *T = P->T, *psi = P->psi; const double *h = P->hs;
    const double *y0 = P->Y[0], *y1 = P->Y[1], *y2 = P->Y[2], *y3 = P->Y[3], *y4 = P->Y[4];
    const double w0 = WMOL[0], w1 = WMOL[1], w2 = WMOL[2], w3 = WMOL[3], w4 = WMOL[4];
    forN(P, P->N, [=] __device__(long i) {
        const double t = TREF + h[i] / CP;
        T[i] = t;
        psi[i] = 1.0 / (RR * t * ((((y0[i] / w0 + y1[i] / w1) + y2[i] / w2) + y3[i] / w3) + y4[i] / w4));
    });
}
static void mul(ffm_plume *P, double *o, const double *a, const double *b, long n) { forN(P, n, [=] __device__(long i) { o[i] = a[i] * b[i]; }); }

// boundary zero-gradient copy of a cell field
static void zg(ffm_plume *P, double *ob, const double *vf)
{ const int *fc = ffm_mesh_bcells(P->mesh); forN(P, P->B, [=] __device__(long k) { ob[k] = vf[fc[k]]; }); }

// dynamic part of the mixed BCs: inletOutlet / pressureInletOutletVelocity value fraction f = 1 - pos0(phi_b)
static void bc_update_f(ffm_plume *P, double *f, const double *fStatic)
{ const double *pb = P->phib; forN(P, P->B, [=] __device__(long k) { f[k] = fStatic[k] < 0 ? 1.0 - (pb[k] >= 0 ? 1.0 : 0.0) : fStatic[k]; }); }

static int solve_named(ffm_plume *P, const char *name, int solver, int pre, double tol, double relTol, const double *d,
                       const double *up, const double *lo, double *psi, const double *src, bool sameOffDiag = false, bool keepSolver = false)
{
    // zero-copy: the driver's coefficient arrays stay untouched until the solve has returned
    FFM_TRY(ffm_ldu_bind_coeffs_native_d(P->A, d, up, lo, sameOffDiag ? 1 : 0));
    SolveLog L; memset(&L, 0, sizeof(L)); strncpy(L.name, name, sizeof(L.name) - 1);
    if (P->tight) { tol = 1e-13; relTol = 0.0; }
    int maxIter = 1000;
    if (P->stecklerSolvers && solver == FFM_PBICGSTAB && !keepSolver) { solver = FFM_SMOOTH; pre = FFM_SYMGS; maxIter = P->tight ? 1000 : 10; }
    FFM_TRY(ffm_solve_d(P->A, solver, pre, tol, relTol, 0, maxIter, 1, psi, src, &L.perf));
    P->log.push_back(L);
    return FFM_OK;
}

// n systems with the off-diagonal coefficients up / lo (the components of U; the species under the common limiter): ffm_solve_multi_d
static int solve_named_multi(ffm_plume *P, int n, const char *const *names, double tol, const double *const *d, const double *up, const double *lo,
                             double *const *psi, const double *const *src)
{
    if (P->stecklerSolvers || n < 2) {
        for (int i = 0; i < n; i++) FFM_TRY(solve_named(P, names[i], FFM_PBICGSTAB, FFM_DILU, tol, 0.0, d[i], up, lo, psi[i], src[i], i > 0));
        return FFM_OK;
    }
    double relTol = 0.0;
    if (P->tight) { tol = 1e-13; relTol = 0.0; }
    std::vector<ffm_perf> pf(n);
    FFM_TRY(ffm_solve_multi_d(P->A, n, FFM_PBICGSTAB, FFM_DILU, tol, relTol, 0, 1000, d, up, lo, psi, src, pf.data()));
    for (int i = 0; i < n; i++) { SolveLog L; memset(&L, 0, sizeof(L)); strncpy(L.name, names[i], sizeof(L.name) - 1); L.perf = pf[i]; P->log.push_back(L); }
    return FFM_OK;
}

// ---- boundary values of U from its mixed BC (per component) -----------------------------------
static int U_boundary(ffm_plume *P, double *Ub[3])
{
    for (int c = 0; c < 3; c++) FFM_TRY(ffm_bc_values(P->mesh, P->fU[c], P->refU[c], P->zeroB, P->U[c], Ub[c]));
    return FFM_OK;
}

static int update_bcs(ffm_plume *P)
{
    for (int c = 0; c < 3; c++) bc_update_f(P, P->fU[c], P->fStaticU[c]);
    bc_update_f(P, P->fS, P->fStaticS);
    bc_update_f(P, P->fH, P->fStaticH);
    return FFM_OK;
}

// p_rgh BC: fixedFluxPressure gradient on inlet/floor, prghTotalHydrostaticPressure value on top/sides
static int bc_p_rgh(ffm_plume *P, const double *grad /*[B] or null*/, double *Ub[3], const double *rhob)
{
    const double *kind = P->kind_d, *pb = P->phib, *phb = P->ph_rgh_b;
    const double *u0 = Ub[0], *u1 = Ub[1], *u2 = Ub[2];
    double *f = P->fP, *ref = P->refP, *g = P->gradP;
    forN(P, P->B, [=] __device__(long k) {
        if (kind[k] < 1.5) { f[k] = 0.0; ref[k] = 0.0; g[k] = grad ? grad[k] : 0.0; }
        else {
            f[k] = 1.0; g[k] = 0.0;
            ref[k] = phb[k] - 0.5 * rhob[k] * (1.0 - (pb[k] >= 0 ? 1.0 : 0.0)) * ((u0[k] * u0[k] + u1[k] * u1[k]) + u2[k] * u2[k]);
        }
    });
    return FFM_OK;
}

static int rho_eqn(ffm_plume *P)
{   // fvm::ddt(rho) + fvc::div(phi) == 0  -> diagonal: rho = (rdt*rho0*V - V*div(phi))/(rdt*V)
    double *div = P->wN[0];
    FFM_TRY(ffm_fvc_surface_integrate(P->mesh, P->phi, P->phib, div));
    const double *V = ffm_mesh_geom(P->mesh, 0), *rho0 = P->rho0; double *rho = P->rho; const double rdt = P->rdt;
    forN(P, P->nOwn, [=] __device__(long i) { rho[i] = (rdt * rho0[i] * V[i] - V[i] * div[i]) / (rdt * V[i]); });
    return HX(P, rho);
}

static int hydrostatic_init(ffm_plume *P)
{
    ffm_mesh *m = P->mesh; const int N = P->N, B = P->B;
    double *ph = P->ph_rgh, *rhof = P->wF[0], *sg = P->wF[1], *phig = P->wF[2], *rhob = P->wB[0];
    const double *gh = P->gh, *ghf = P->ghf, *magSf = ffm_mesh_geom(m, 1);
    double *p = P->p, *rho = P->rho; const double *psi = P->psi;
    // top fixedValue 0, everything else fixedFluxPressure with zero gradient
    double *fTop = P->wB[1]; const double *kind = P->kind_d;
    forN(P, B, [=] __device__(long k) { fTop[k] = (kind[k] > 1.5 && kind[k] < 2.5) ? 1.0 : 0.0; });
    forN(P, N, [=] __device__(long i) { p[i] = ph[i] + rho[i] * gh[i] + PREF; });
    standin_thermo(P); mul(P, rho, psi, p, N);
    const long nNat = P->nNat;
    for (int corr = 0; corr < 5; corr++) {
        FFM_TRY(ffm_fvc_interpolate(m, nullptr, rho, rhof));
        zg(P, rhob, rho);
        FFM_TRY(ffm_fvc_snGrad(m, rho, sg));
        forN(P, nNat, [=] __device__(long e) { phig[e] = -rhof[e] * ghf[e] * sg[e] * magSf[e]; });
        FFM_TRY(ffm_fvm_transport(m, 0.0, nullptr, nullptr, nullptr, rhof, +1, P->diag, P->upper, P->lower));
        FFM_TRY(ffm_fvm_boundary_coeffs(m, nullptr, rhob, +1, fTop, P->zeroB, P->zeroB, P->ic[0], P->bc[0]));
        FFM_TRY(ffm_fvc_surface_integrate(m, phig, P->zeroB, P->wN[0]));
        {
            double *s0 = P->src[0]; const double *dv = P->wN[0], *V = ffm_mesh_geom(m, 0);
            forN(P, N, [=] __device__(long i) { s0[i] = V[i] * dv[i]; });
        }
        FFM_TRY(ffm_fvm_add_boundary(m, P->ic[0], P->bc[0], P->diag, P->src[0], nullptr, P->dWork, P->sWork));
        FFM_TRY(solve_named(P, "ph_rgh", FFM_PCG, FFM_DIC, 1e-6, 0.01, P->dWork, P->upper, nullptr, ph, P->sWork));
        FFM_TRY(HX(P, ph));
        forN(P, N, [=] __device__(long i) { p[i] = ph[i] + rho[i] * gh[i] + PREF; });
        standin_thermo(P); mul(P, rho, psi, p, N);
    }
    FFM_TRY(ffm_bc_values(m, fTop, P->zeroB, P->zeroB, ph, P->ph_rgh_b));
    dcopy(P, P->p_rgh, ph, N);
    return FFM_OK;
}

// transport equation of a scalar: ddt(rho,vf) + div(phi,vf) - laplacian(gamma,vf) == su (+ explicit LHS terms in `expl`)
static int scalar_transport(ffm_plume *P, const char *name, int scheme, double *vf, const double *vf0, const double *fBC, const double *ref,
                            const double *gamma_f, const double *gamma_b, const double *su, const double *const *expl, double tol,
                            const double *su2 = nullptr, const double *sp = nullptr, const double *wGiven = nullptr)
{
    ffm_mesh *m = P->mesh; const int N = P->N;
    double *vb = P->wB[2], *gx = P->wN[1], *gy = P->wN[2], *gz = P->wN[3];
    const double *w = wGiven;
    if (!wGiven) {
        FFM_TRY(ffm_bc_values(m, fBC, ref, P->zeroB, vf, vb));
        FFM_TRY(ffm_fvc_grad(m, vf, vb, gx, gy, gz));
        FFM_TRY(HX(P, gx)); FFM_TRY(HX(P, gy)); FFM_TRY(HX(P, gz));
        FFM_TRY(ffm_fv_limited_weights(m, scheme, 1.0, 0.0, 1.0, P->phi, vf, gx, gy, gz, P->wF[3]));
        w = P->wF[3];
    }
    FFM_TRY(ffm_fvm_transport(m, P->rdt, P->rho, P->phi, w, gamma_f, -1, P->diag, P->upper, P->lower));
    FFM_TRY(ffm_fvm_boundary_coeffs(m, P->phib, gamma_b, -1, fBC, ref, P->zeroB, P->ic[0], P->bc[0]));
    // source = rdt*rho0*vf0*V (- V*expl) ; then + boundaryCoeffs + V*su
    const double *V = ffm_mesh_geom(m, 0), *rho0 = P->rho0; double *s = P->src[0]; const double rdt = P->rdt;
    // explicit volume terms on the left-hand side: one `source -= V*term` each, in the order given (fvMatrix + volField)
    if (expl) { const double *e0 = expl[0], *e1 = expl[1], *e2 = expl[2];
                forN(P, N, [=] __device__(long i) { s[i] = ((rdt * rho0[i] * vf0[i] * V[i] - V[i] * e0[i]) - V[i] * e1[i]) - V[i] * e2[i]; }); }
    else forN(P, N, [=] __device__(long i) { s[i] = rdt * rho0[i] * vf0[i] * V[i]; });
    double *s2 = P->wN[4];
    if (su) { forN(P, N, [=] __device__(long i) { s2[i] = s[i] + V[i] * su[i]; }); }
    else s2 = s;
    if (sp) { double *dg = P->diag; forN(P, N, [=] __device__(long i) { dg[i] = dg[i] + V[i] * sp[i]; }); }      // - fvm::Sp(sp, vf) on the RHS
    if (su2) { double *s3 = P->src[1]; const double *sIn = s2; forN(P, N, [=] __device__(long i) { s3[i] = sIn[i] + V[i] * su2[i]; }); s2 = s3; }
    FFM_TRY(ffm_fvm_add_boundary(m, P->ic[0], P->bc[0], P->diag, s2, nullptr, P->dWork, P->sWork));
    FFM_TRY(solve_named(P, name, FFM_PBICGSTAB, FFM_DILU, tol, 0.0, P->dWork, P->upper, P->lower, vf, P->sWork));
    return HX(P, vf);
}

// ---- fvDOM stand-in: radiation->correct() of solver/YEEqn.H:80 -------------------------------------------------------
// per ray i (direction dAve_i, solid angle omega_i; fvDOM.C:55-90, radiativeIntensityRay.C:126-143):
//   fvm::div(Ji, Ii) + fvm::Sp(k*omega, Ii) == 1/pi*omega*(k*sigma*T^4),  Ji = dAve & Sf, div scheme upwind
// (radiativeIntensityRay.C:267-322), inflow faces at the ambient black-body intensity, outflow zero-gradient; then
// G = sum Ii*omega (fvDOM::updateG).  Constant k, no scattering, no coupling back into the enthalpy equation.
constexpr double SIGMA_SB = 5.670367e-8;
extern "C" int ffm_reduce_sum(ffm_ctx *, const double *, long, double *);
// RadFraction of constRadFractionEmission::ECont with radScaling (reference lib/thermophysicalModels/radiation/submodels/
// absorptionEmissionModel/constRadFractionEmission/constRadFractionEmission.C): both patch lists name the burner
// (cases/steckler/constant/radiationProperties:44-52) -> mlr1 = mlr2 = -gSum(phi_burner)
static int plume_rad_fraction(ffm_plume *P, double *out)
{
    const double *kind = P->kind_d, *pb = P->phib; double *t = P->wB[0];
    forN(P, P->B, [=] __device__(long k) { t[k] = kind[k] < 0.5 ? pb[k] : 0.0; });
    double sum = 0.0;
    FFM_TRY(ffm_reduce_sum(P->ctx, t, P->B, &sum));          // gSum: every rank takes part, also one without boundary faces
    const double mlr = -sum, e1 = P->Ehrr1, e2 = P->Ehrr2;
    *out = std::max(std::min(e1, e2), (mlr * e1 + mlr * e2) / std::max(1e-15, mlr + mlr));
    return FFM_OK;
}
static int radiation_correct(ffm_plume *P)
{
    ffm_mesh *m = P->mesh; const int N = P->N, B = P->B; const long nNat = P->nNat;
    const double *V = ffm_mesh_geom(m, 0);
    const double *sx = ffm_mesh_geom(m, 9), *sy = ffm_mesh_geom(m, 10), *sz = ffm_mesh_geom(m, 11);
    const double *bx = ffm_mesh_geom(m, 6), *by = ffm_mesh_geom(m, 7), *bz = ffm_mesh_geom(m, 8);
    const double Ib = SIGMA_SB * ((TREF * TREF) * (TREF * TREF)) / M_PI;
    double *G = P->G, *J = P->radJ, *w = P->radW, *Jb = P->radJb, *f = P->radF, *ref = P->radRef, *su = P->radSrc;
    const double *T = P->T;
    forN(P, N, [=] __device__(long c) { G[c] = 0.0; });
    forN(P, B, [=] __device__(long k) { ref[k] = Ib; });
    const double KA = P->radA;
    const double *Ee = nullptr;
    if (P->radCoupled) {      // absorptionEmission->ECont(): E = RadFraction*Qdot (constRadFractionEmission.C, radScaling)
        double frac = 0.0;
        FFM_TRY(plume_rad_fraction(P, &frac));
        double *E = P->radE; const double *Qd = P->wN[9];
        forN(P, N, [=] __device__(long c) { E[c] = frac * Qd[c]; });
        Ee = E;
    }
    const int nRay = (int)P->rayOmega.size();
    for (int i = 0; i < nRay; i++) {
        const double d0 = P->rayD[3 * i], d1 = P->rayD[3 * i + 1], d2 = P->rayD[3 * i + 2], omega = P->rayOmega[i];
        forN(P, nNat, [=] __device__(long e) { const double j = (d0 * sx[e] + d1 * sy[e]) + d2 * sz[e]; J[e] = j; w[e] = j >= 0 ? 1.0 : 0.0; });
        forN(P, B, [=] __device__(long k) { const double j = (d0 * bx[k] + d1 * by[k]) + d2 * bz[k]; Jb[k] = j; f[k] = 1.0 - (j >= 0 ? 1.0 : 0.0); });
        FFM_TRY(ffm_fvm_transport(m, 0.0, nullptr, J, w, nullptr, -1, P->diag, P->upper, P->lower));
        FFM_TRY(ffm_fvm_boundary_coeffs(m, Jb, nullptr, -1, f, ref, P->zeroB, P->ic[0], P->bc[0]));
        double *dg = P->diag; const double kO = KA * omega, cS = 1.0 / M_PI * omega, kS = KA * SIGMA_SB;
        forN(P, N, [=] __device__(long c) {
            dg[c] = dg[c] + V[c] * kO;
            const double t = T[c];
            // 1/pi*omega*(k sigma T^4 [+ E/4]) (radiativeIntensityRay.C:286-300)
            su[c] = Ee ? V[c] * (cS * (kS * ((t * t) * (t * t)) + Ee[c] / 4.0)) : V[c] * (cS * (kS * ((t * t) * (t * t))));
        });
        FFM_TRY(ffm_fvm_add_boundary(m, P->ic[0], P->bc[0], P->diag, su, nullptr, P->dWork, P->sWork));
        char nm[16]; snprintf(nm, sizeof(nm), "I%d", i);
        // the axis whose direction component has the minority sign; none when all three agree (the cell order is then already
        // an upwind or a downwind order of the ray and DILU is exact)
        int flip = -1;
        if (P->radOrdered) {
            const int neg = (d0 < 0) + (d1 < 0) + (d2 < 0);
            if (neg == 1) flip = d0 < 0 ? 0 : d1 < 0 ? 1 : 2;
            else if (neg == 2) flip = d0 >= 0 ? 0 : d1 >= 0 ? 1 : 2;
        }
        if (flip < 0) {
            FFM_TRY(solve_named(P, nm, FFM_PBICGSTAB, FFM_DILU, 1e-4, 0.0, P->dWork, P->upper, P->lower, P->I[i], P->sWork, false, true));
        } else {
            // the same system with the cells renamed by the flip of that axis: same sparsity, triangular
            const int *cm = P->radCm[flip], *fm = P->radFm[flip];
            const double *dW = P->dWork, *sW = P->sWork, *up = P->upper, *lo = P->lower; double *Ii = P->I[i];
            double *dB = P->radDB, *sB = P->radSB, *pB = P->radPsiB, *uB = P->radUB, *lB = P->radLB;
            forN(P, N, [=] __device__(long c) { const int s = cm[c]; dB[c] = dW[s]; sB[c] = sW[s]; pB[c] = Ii[s]; });
            forN(P, nNat, [=] __device__(long e) {
                const int t = fm[e];
                if (t >= 0) { uB[e] = up[t]; lB[e] = lo[t]; } else { uB[e] = lo[~t]; lB[e] = up[~t]; }
            });
            FFM_TRY(solve_named(P, nm, FFM_PBICGSTAB, FFM_DILU, 1e-4, 0.0, dB, uB, lB, pB, sB, false, true));
            forN(P, N, [=] __device__(long c) { Ii[cm[c]] = pB[c]; });
        }
        FFM_TRY(HX(P, P->I[i]));
        const double *Ii = P->I[i];
        forN(P, N, [=] __device__(long c) { G[c] = G[c] + Ii[c] * omega; });
    }
    P->radHaveG = true;
    return FFM_OK;
}

static int p_corrector(ffm_plume *P, bool final)
{
    ffm_mesh *m = P->mesh; const int N = P->N, B = P->B; const long nNat = P->nNat; const double rdt = P->rdt;
    const double *V = ffm_mesh_geom(m, 0), *magSf = ffm_mesh_geom(m, 1), *bMag = ffm_mesh_geom(m, 4);
    const double *bSx = ffm_mesh_geom(m, 6), *bSy = ffm_mesh_geom(m, 7), *bSz = ffm_mesh_geom(m, 8);
    const double *Sx = ffm_mesh_geom(m, 9), *Sy = ffm_mesh_geom(m, 10), *Sz = ffm_mesh_geom(m, 11), *wlin = ffm_mesh_geom(m, 3);
    double *rho = P->rho; const double *psi = P->psi; double *p = P->p;
    double *rAU = P->wN[0], *rhorAU = P->wN[1], *HbyA[3] = {P->wN[2], P->wN[3], P->wN[4]};
    FFM_TRY(ffm_fvm_A(m, 3, P->Udiag, P->Uic[0], P->Uic[1], P->Uic[2], rAU));
    if (P->fused && P->N == P->nOwn) {
        // rho = thermo.rho(); rAU = 1/A; rhorAU = rho*rAU in one pass (single block: no ghost refresh in between)
        forN(P, N, [=] __device__(long i) { const double r = psi[i] * p[i], a = 1.0 / rAU[i]; rho[i] = r; rAU[i] = a; rhorAU[i] = r * a; });
    } else {
    mul(P, rho, psi, p, N);                                                        // rho = thermo.rho()
    forN(P, P->nOwn, [=] __device__(long i) { rAU[i] = 1.0 / rAU[i]; });
    FFM_TRY(HX(P, rAU));
    forN(P, N, [=] __device__(long i) { rhorAU[i] = rho[i] * rAU[i]; });
    }
    double *rhorAUf = P->wF[0], *rhorAUfb = P->wB[0];
    FFM_TRY(ffm_fvc_interpolate(m, nullptr, rhorAU, rhorAUf));
    zg(P, rhorAUfb, rhorAU);
    if (P->fused) {
        const double *srcs[3] = {P->Usrc[0], P->Usrc[1], P->Usrc[2]}, *ics[3] = {P->Uic[0], P->Uic[1], P->Uic[2]}, *bcs[3] = {P->Ubc[0], P->Ubc[1], P->Ubc[2]};
        const double *us[3] = {P->U[0], P->U[1], P->U[2]};
        FFM_TRY(ffm_fvm_HbyA3(m, P->Uupper, P->Ulower, srcs, ics, bcs, us, rAU, HbyA));
        for (int c = 0; c < 3; c++) FFM_TRY(HX(P, HbyA[c]));
    } else
    for (int c = 0; c < 3; c++) {
        FFM_TRY(ffm_fvm_H(m, 3, c, P->Uupper, P->Ulower, P->Usrc[c], P->Uic[0], P->Uic[1], P->Uic[2], P->Ubc[c], P->U[c], HbyA[c]));
        double *Hc = HbyA[c];
        forN(P, P->nOwn, [=] __device__(long i) { Hc[i] = rAU[i] * Hc[i]; });
        FFM_TRY(HX(P, Hc));
    }
    FFM_TRY(update_bcs(P));
    double *Ub[3] = {P->wB[1], P->wB[2], P->wB[3]};
    FFM_TRY(U_boundary(P, Ub));
    double *rhob = P->wB[4];
    zg(P, rhob, rho);
    // phig = -rhorAUf*ghf*snGrad(rho)*magSf
    double *sg = P->wF[1], *phig = P->wF[2];
    const double *ghf = P->ghf;
    if (P->fused) FFM_TRY(ffm_pc_phig(m, rhorAUf, ghf, rho, phig));
    else {
    FFM_TRY(ffm_fvc_snGrad(m, rho, sg));
    forN(P, nNat, [=] __device__(long e) { phig[e] = -rhorAUf[e] * ghf[e] * sg[e] * magSf[e]; });
    }
    // fvc::flux(rho*HbyA): interior by linear interpolation; boundary rho_b*HbyA_b.Sf with constrainHbyA
    double *phiHbyA = P->wF[3], *phiHbyAb = P->wB[5];
    if (P->fused) { /* formed below, together with the ddtCorr and phig terms (ffm_pc_phiHbyA) */ }
    else {
        double *rH[3] = {P->wN[5], P->wN[6], P->wN[7]};
        for (int c = 0; c < 3; c++) mul(P, rH[c], rho, HbyA[c], N);
        FFM_TRY(ffm_fvc_flux(m, rH[0], rH[1], rH[2], phiHbyA));
    }
    {
        const int *fc = ffm_mesh_bcells(m); const double *kind = P->kind_d;
        const double *h0 = HbyA[0], *h1 = HbyA[1], *h2 = HbyA[2], *u0 = Ub[0], *u1 = Ub[1], *u2 = Ub[2];
        forN(P, B, [=] __device__(long k) {
            const bool fixed = kind[k] < 1.5; const int c = fc[k];
            const double a0 = fixed ? u0[k] : h0[c], a1 = fixed ? u1[k] : h1[c], a2 = fixed ? u2[k] : h2[c];
            phiHbyAb[k] = (rhob[k] * a0 * bSx[k] + rhob[k] * a1 * bSy[k]) + rhob[k] * a2 * bSz[k];
        });
    }
    // + rhorAUf*ddtCorr(rho,U,phi) + phig
    {
        double *dc = P->ddtCorrF;
        if (!P->ddtCorrValid) {         // old-time fields only: evaluated in the first corrector of a step, reused by the second
            double *fl0 = P->wF[4];
            if (P->fused) FFM_TRY(ffm_fvc_flux_rho(m, P->rho0, P->U0[0], P->U0[1], P->U0[2], fl0));
            else {
                double *rU0[3] = {P->wN[8], P->wN[9], P->wN[10]};
                for (int c = 0; c < 3; c++) mul(P, rU0[c], P->rho0, P->U0[c], N);
                FFM_TRY(ffm_fvc_flux(m, rU0[0], rU0[1], rU0[2], fl0));
            }
            const double *phi0 = P->phi0;
            forN(P, nNat, [=] __device__(long e) {
                const double phiCorr = phi0[e] - fl0[e];
                const double coeff = 1.0 - fmin(fabs(phiCorr) / (fabs(phi0[e]) + 1e-15), 1.0);
                dc[e] = coeff * rdt * phiCorr;
            });
            P->ddtCorrValid = true;
        }
        if (P->fused) FFM_TRY(ffm_pc_phiHbyA(m, rho, HbyA[0], HbyA[1], HbyA[2], rhorAUf, dc, phig, phiHbyA));
        else forN(P, nNat, [=] __device__(long e) { phiHbyA[e] = (phiHbyA[e] + rhorAUf[e] * dc[e]) + phig[e]; });
        (void)Sx; (void)Sy; (void)Sz; (void)wlin;
    }
    // constrainPressure: gradient on fixedFluxPressure patches
    double *grads = P->wB[6];
    {
        const double *u0 = Ub[0], *u1 = Ub[1], *u2 = Ub[2];
        forN(P, B, [=] __device__(long k) {
            grads[k] = (phiHbyAb[k] - rhob[k] * ((bSx[k] * u0[k] + bSy[k] * u1[k]) + bSz[k] * u2[k])) / (bMag[k] * rhorAUfb[k]);
        });
    }
    FFM_TRY(bc_p_rgh(P, grads, Ub, rhob));
    // p_rghEqn = fvm::ddt(psi,p_rgh) + fvc::ddt(psi,rho)*gh + fvc::ddt(psi)*pRef + fvc::div(phiHbyA) - fvm::laplacian(rhorAUf,p_rgh)
    if (P->fused) {
        FFM_TRY(ffm_fvm_boundary_coeffs(m, nullptr, rhorAUfb, -1, P->fP, P->refP, P->gradP, P->ic[0], P->bc[0]));
        FFM_TRY(ffm_fvm_pressure_eqn(m, rdt, psi, P->psi0, P->p_rgh0, rho, P->rho0, P->gh, PREF, rhorAUf, phiHbyA, phiHbyAb, P->ic[0], P->bc[0],
                                     P->upper, P->lower, P->dWork, P->sWork));
    } else {
    FFM_TRY(ffm_fvm_transport(m, rdt, psi, nullptr, nullptr, rhorAUf, -1, P->diag, P->upper, P->lower));
    FFM_TRY(ffm_fvm_boundary_coeffs(m, nullptr, rhorAUfb, -1, P->fP, P->refP, P->gradP, P->ic[0], P->bc[0]));
    double *div = P->wN[8];
    FFM_TRY(ffm_fvc_surface_integrate(m, phiHbyA, phiHbyAb, div));
    {
        double *s = P->src[0]; const double *psi0 = P->psi0, *prgh0 = P->p_rgh0, *rho0 = P->rho0, *gh = P->gh;
        forN(P, N, [=] __device__(long i) {
            // fvc::ddt(psi,rho)*gh, fvc::ddt(psi)*pRef, fvc::div(phiHbyA): one source update each (solver/pEqn.H:30-33)
            s[i] = ((rdt * psi0[i] * prgh0[i] * V[i] - V[i] * (rdt * (psi[i] * rho[i] - psi0[i] * rho0[i]) * gh[i]))
                    - V[i] * (rdt * (psi[i] - psi0[i]) * PREF)) - V[i] * div[i];
        });
    }
    FFM_TRY(ffm_fvm_add_boundary(m, P->ic[0], P->bc[0], P->diag, P->src[0], nullptr, P->dWork, P->sWork));
    }
    FFM_TRY(solve_named(P, "p_rgh", FFM_PCG, FFM_DIC, 1e-6, final ? 0.0 : 0.01, P->dWork, P->upper, nullptr, P->p_rgh, P->sWork));
    FFM_TRY(HX(P, P->p_rgh));
    // phi = phiHbyA + p_rghEqn.flux(); U = HbyA + rAU*reconstruct((flux + phig)/rhorAUf)
    double *fl = P->wF[4], *flb = P->wB[7];
    if (P->fused) {
        FFM_TRY(ffm_fvm_flux(m, P->upper, P->lower, P->ic[0], P->bc[0], P->p_rgh, nullptr, flb));
        FFM_TRY(ffm_pc_flux(m, P->upper, P->lower, P->p_rgh, phiHbyA, phig, rhorAUf, fl, P->phi, P->wF[5]));
    } else
    FFM_TRY(ffm_fvm_flux(m, P->upper, P->lower, P->ic[0], P->bc[0], P->p_rgh, fl, flb));
    {
        double *phi = P->phi, *phib = P->phib, *t = P->wF[5], *tb = P->wB[6];
        if (!P->fused) forN(P, nNat, [=] __device__(long e) { phi[e] = phiHbyA[e] + fl[e]; t[e] = rhorAUf[e] != 0.0 ? (fl[e] + phig[e]) / rhorAUf[e] : 0.0; });
        forN(P, B, [=] __device__(long k) { phib[k] = phiHbyAb[k] + flb[k]; tb[k] = flb[k] / rhorAUfb[k]; });
        double *rx = P->wN[5], *ry = P->wN[6], *rz = P->wN[7];
        FFM_TRY(ffm_fvc_reconstruct(m, t, tb, rx, ry, rz));
        double *U0 = P->U[0], *U1 = P->U[1], *U2 = P->U[2]; const double *h0 = HbyA[0], *h1 = HbyA[1], *h2 = HbyA[2];
        double *K = P->K, *dpdt = P->dpdt; const double *p_rgh = P->p_rgh, *gh = P->gh, *p0 = P->p0;
        if (P->fused && P->N == P->nOwn) {
            // U = HbyA + rAU*reconstruct(...), K = 0.5 magSqr(U), p = p_rgh + rho*gh + pRef, dpdt = fvc::ddt(p) in one pass over the cells
            // (single block; rhoEqn.H, which follows p in solver/pEqn.H:46-48, reads neither)
            forN(P, N, [=] __device__(long i) {
                const double a = h0[i] + rAU[i] * rx[i], b = h1[i] + rAU[i] * ry[i], c = h2[i] + rAU[i] * rz[i];
                U0[i] = a; U1[i] = b; U2[i] = c;
                K[i] = 0.5 * ((a * a + b * b) + c * c);
                const double pp = p_rgh[i] + rho[i] * gh[i] + PREF;
                p[i] = pp; dpdt[i] = rdt * (pp - p0[i]);
            });
            FFM_TRY(rho_eqn(P));
        } else {
        forN(P, P->nOwn, [=] __device__(long i) {
            const double a = h0[i] + rAU[i] * rx[i], b = h1[i] + rAU[i] * ry[i], c = h2[i] + rAU[i] * rz[i];
            U0[i] = a; U1[i] = b; U2[i] = c;
        });
        FFM_TRY(HX(P, U0)); FFM_TRY(HX(P, U1)); FFM_TRY(HX(P, U2));
        forN(P, N, [=] __device__(long i) { p[i] = p_rgh[i] + rho[i] * gh[i] + PREF; });
        FFM_TRY(rho_eqn(P));
        forN(P, N, [=] __device__(long i) {
            K[i] = 0.5 * ((U0[i] * U0[i] + U1[i] * U1[i]) + U2[i] * U2[i]);
            dpdt[i] = rdt * (p[i] - p0[i]);
        });
        }
    }
    return FFM_OK;
}

extern "C" int ffm_plume_step(ffm_plume *P)
{
    if (!P) return FFM_ERR_ARG;
    ffm_mesh *m = P->mesh; const int N = P->N, B = P->B; const long nNat = P->nNat; const double rdt = P->rdt;
    const double *V = ffm_mesh_geom(m, 0), *magSf = ffm_mesh_geom(m, 1);
    P->log.clear();
    P->ddtCorrValid = false;
    // oldTime fields
    dcopy(P, P->rho0, P->rho, N); dcopy(P, P->hs0, P->hs, N); dcopy(P, P->K0, P->K, N); dcopy(P, P->p0, P->p, N);
    dcopy(P, P->psi0, P->psi, N); dcopy(P, P->p_rgh0, P->p_rgh, N); dcopy(P, P->phi0, P->phi, nNat); dcopy(P, P->phib0, P->phib, B);
    for (int c = 0; c < 3; c++) dcopy(P, P->U0[c], P->U[c], N);
    for (int i = 0; i < NSP; i++) dcopy(P, P->Y0[i], P->Y[i], N);
    FFM_TRY(rho_eqn(P));
    // ---------------- UEqn.H
    FFM_TRY(update_bcs(P));
    double *Ub[3] = {P->wB[1], P->wB[2], P->wB[3]};
    FFM_TRY(U_boundary(P, Ub));
    // div(phi,U) Gauss LUST grad(U) (cases/steckler/system/fvSchemes:32): LUST weights for the implicit part; the explicit
    // correction is added to the source below
    double *gx = P->wN[1], *gy = P->wN[2], *gz = P->wN[3], *wU = P->wF[3];
    FFM_TRY(ffm_fv_limited_weights(m, 4, 1.0, 0.0, 1.0, P->phi, nullptr, nullptr, nullptr, nullptr, wU));
    double *muf = P->wF[0], *mub = P->wB[4];
    forN(P, nNat, [=] __device__(long e) { muf[e] = MU; });
    forN(P, B, [=] __device__(long k) { mub[k] = MU; });
    FFM_TRY(ffm_fvm_transport(m, rdt, P->rho, P->phi, wU, muf, -1, P->Udiag, P->Uupper, P->Ulower));
    // reconstruct((-ghf*snGrad(rho) - snGrad(p_rgh))*magSf)
    double *sgr = P->wF[1], *sgp = P->wF[2], *t = P->wF[4], *tb = P->wB[5], *rhob = P->wB[6], *pb = P->wB[7];
    if (!P->fused) FFM_TRY(ffm_fvc_snGrad(m, P->rho, sgr));
    zg(P, rhob, P->rho);
    FFM_TRY(bc_p_rgh(P, nullptr, Ub, rhob));
    FFM_TRY(ffm_bc_values(m, P->fP, P->refP, P->gradP, P->p_rgh, pb));
    if (!P->fused) FFM_TRY(ffm_fvc_snGrad(m, P->p_rgh, sgp));
    FFM_TRY(ffm_fvc_snGrad_b(m, P->p_rgh, pb, tb));
    {
        const double *ghf = P->ghf, *bMag = ffm_mesh_geom(m, 4);
        if (P->fused) FFM_TRY(ffm_ue_buoyancy_flux(m, ghf, P->rho, P->p_rgh, t));
        else forN(P, nNat, [=] __device__(long e) { t[e] = (-ghf[e] * sgr[e] - sgp[e]) * magSf[e]; });
        forN(P, B, [=] __device__(long k) { tb[k] = -tb[k] * bMag[k]; });
    }
    double *rx = P->wN[5], *ry = P->wN[6], *rz = P->wN[7];
    FFM_TRY(ffm_fvc_reconstruct(m, t, tb, rx, ry, rz));
    double *rec[3] = {rx, ry, rz};
    if (P->fused) {
        // one gradient pass for the three components, one pass for the LUST correction + the ddt source (ffm_fused.hip)
        for (int c = 0; c < 3; c++) FFM_TRY(ffm_fvm_boundary_coeffs(m, P->phib, mub, -1, P->fU[c], P->refU[c], P->zeroB, P->Uic[c], P->Ubc[c]));
        const double *uf[3] = {P->U[0], P->U[1], P->U[2]}, *ub[3] = {Ub[0], Ub[1], Ub[2]}, *u0[3] = {P->U0[0], P->U0[1], P->U0[2]};
        double *ggx[3] = {P->gM[0][0], P->gM[1][0], P->gM[2][0]}, *ggy[3] = {P->gM[0][1], P->gM[1][1], P->gM[2][1]}, *ggz[3] = {P->gM[0][2], P->gM[1][2], P->gM[2][2]};
        FFM_TRY(ffm_fvc_grad_multi(m, 3, uf, ub, ggx, ggy, ggz));
        for (int c = 0; c < 3; c++) { FFM_TRY(HX(P, ggx[c])); FFM_TRY(HX(P, ggy[c])); FFM_TRY(HX(P, ggz[c])); }
        FFM_TRY(ffm_fvm_lust_source3(m, rdt, P->phi, P->rho0, u0, ggx, ggy, ggz, P->Usrc));
    } else
    for (int c = 0; c < 3; c++) {
        FFM_TRY(ffm_fvm_boundary_coeffs(m, P->phib, mub, -1, P->fU[c], P->refU[c], P->zeroB, P->Uic[c], P->Ubc[c]));
        // gaussConvectionScheme::fvmDiv with a corrected() scheme: fvm += fvc::surfaceIntegrate(phi*LUST::correction(U_c))
        double *corr = P->wF[4], *divc = P->wN[4];
        FFM_TRY(ffm_fvc_grad(m, P->U[c], Ub[c], gx, gy, gz));
        FFM_TRY(HX(P, gx)); FFM_TRY(HX(P, gy)); FFM_TRY(HX(P, gz));
        FFM_TRY(ffm_fv_lust_correction(m, P->phi, gx, gy, gz, corr));
        { const double *phi = P->phi; forN(P, nNat, [=] __device__(long e) { corr[e] = phi[e] * corr[e]; }); }
        FFM_TRY(ffm_fvc_surface_integrate(m, corr, P->zeroB, divc));
        double *s = P->Usrc[c]; const double *rho0 = P->rho0, *u0 = P->U0[c];
        forN(P, N, [=] __device__(long i) { s[i] = rdt * rho0[i] * u0[i] * V[i] - V[i] * divc[i]; });
    }
    {
        // fvMatrix::solveSegregated: the three components share the face coefficients -- one lock-step solve (ffm_solve_multi_d)
        const char *nm[3] = {"Ux", "Uy", "Uz"};
        const double *dd[3], *ss[3]; double *pp[3];
        for (int c = 0; c < 3; c++) {
            if (!P->UdW[c]) { P->UdW[c] = dalloc(P, N); P->UsW[c] = dalloc(P, N); if (!P->UdW[c] || !P->UsW[c]) return FFM_ERR_HIP; }
            FFM_TRY(ffm_fvm_add_boundary(m, P->Uic[c], P->Ubc[c], P->Udiag, P->Usrc[c], rec[c], P->UdW[c], P->UsW[c]));
            dd[c] = P->UdW[c]; ss[c] = P->UsW[c]; pp[c] = P->U[c];
        }
        FFM_TRY(solve_named_multi(P, 3, nm, 1e-6, dd, P->Uupper, P->Ulower, pp, ss));
        for (int c = 0; c < 3; c++) FFM_TRY(HX(P, P->U[c]));
    }
    {
        double *K = P->K; const double *U0 = P->U[0], *U1 = P->U[1], *U2 = P->U[2];
        forN(P, N, [=] __device__(long i) { K[i] = 0.5 * ((U0[i] * U0[i] + U1[i] * U1[i]) + U2[i] * U2[i]); });
    }
    { const char *st = getenv("FFM_PLUME_STOP"); if (st && atoi(st) == 1) { PL_HIP(hipStreamSynchronize(P->ctx->stream)); return FFM_OK; } }
    // ---------------- YEEqn.H
    double *af = P->wF[0], *afb = P->wB[4];
    forN(P, nNat, [=] __device__(long e) { af[e] = 0.5 * (MU / PR) + (1.0 - 0.5) * (MU / PR); });
    forN(P, B, [=] __device__(long k) { afb[k] = MU / PR; });
    double *wFuel = P->wN[8], *Qdot = P->wN[9], *Yt = P->wN[10], *su = P->wN[11];
    {
        const double *rho = P->rho, *fuel = P->Y[2], *o2 = P->Y[0];
        forN(P, N, [=] __device__(long i) { const double w = rho[i] * fmin(fuel[i], o2[i] / S_O2) / TAU; wFuel[i] = w; Qdot[i] = w * HC; Yt[i] = 0.0; });
    }
    if (P->mvSelection && P->mvOverride) P->mvOverride = false;      // weights of this step were handed in: wMv stays as uploaded
    else if (P->mvSelection) {
        // ---- mvConvection: the common limiter over the five species and h, from the fields as they are now
        int sp4[NSP - 1], n4 = 0;
        for (int i = 0; i < NSP; i++) if (i != INERT) sp4[n4++] = i;
        double *Nb = P->wB[1], *hb = P->wB[3];
        for (int j = 0; j < n4; j++) FFM_TRY(ffm_bc_values(m, P->fS, P->refY[sp4[j]], P->zeroB, P->Y[sp4[j]], P->spB[j]));
        {   // the inert specie's patch values: Y[inertIndex] == 1 - Yt; .max(0) on the patch faces too
            const double *b0 = P->spB[0], *b1 = P->spB[1], *b2 = P->spB[2], *b3 = P->spB[3];
            forN(P, B, [=] __device__(long k) { const double t = ((fmax(b0[k], 0.0) + fmax(b1[k], 0.0)) + fmax(b2[k], 0.0)) + fmax(b3[k], 0.0); Nb[k] = fmax(1.0 - t, 0.0); });
        }
        FFM_TRY(ffm_bc_values(m, P->fH, P->refH, P->zeroB, P->hs, hb));
        int rcTiled = FFM_ERR_UNSUPPORTED;
        if (P->fused) {
            // gradients of the six fields + the common limiter in ONE pass with the cell values staged through LDS on the tile numbering
            // (ffm_fused.hip: k_mv_tile; bit for bit the two gradient passes + ffm_fv_multivariate_weights below)
            const double *af6[6] = {P->hs, P->Y[sp4[0]], P->Y[sp4[1]], P->Y[sp4[2]], P->Y[sp4[3]], P->Y[INERT]};
            const double *ab6[6] = {hb, P->spB[0], P->spB[1], P->spB[2], P->spB[3], Nb};
            const int sch6[6] = {2, 3, 3, 3, 3, 3};
            rcTiled = ffm_fv_multivariate_weights_tiled(m, 6, sch6, 1.0, 0.0, 1.0, P->phi, af6, ab6, P->wMv);
            if (rcTiled != FFM_OK && rcTiled != FFM_ERR_UNSUPPORTED) return rcTiled;
        }
        if (rcTiled == FFM_OK) { /* done */ }
        else if (P->fused) {
            const double *vf[6], *vb[4], *cgx[6], *cgy[6], *cgz[6]; double *ggx[4], *ggy[4], *ggz[4]; int sch[6];
            // h first, then the transported species, then the inert one (the minimum does not depend on the order)
            const double *vf2[2] = {P->hs, P->Y[INERT]}, *vb2[2] = {hb, Nb};
            double *g2x[2] = {P->mvG[0][0], P->mvG[1][0]}, *g2y[2] = {P->mvG[0][1], P->mvG[1][1]}, *g2z[2] = {P->mvG[0][2], P->mvG[1][2]};
            FFM_TRY(ffm_fvc_grad_multi(m, 2, vf2, vb2, g2x, g2y, g2z));
            for (int j = 0; j < n4; j++) { vf[j] = P->Y[sp4[j]]; vb[j] = P->spB[j]; ggx[j] = P->gM[j][0]; ggy[j] = P->gM[j][1]; ggz[j] = P->gM[j][2]; }
            FFM_TRY(ffm_fvc_grad_multi(m, n4, vf, vb, ggx, ggy, ggz));
            for (int j = 0; j < 2; j++) { FFM_TRY(HX(P, g2x[j])); FFM_TRY(HX(P, g2y[j])); FFM_TRY(HX(P, g2z[j])); }
            for (int j = 0; j < n4; j++) { FFM_TRY(HX(P, ggx[j])); FFM_TRY(HX(P, ggy[j])); FFM_TRY(HX(P, ggz[j])); }
            const double *af6[6] = {P->hs, P->Y[sp4[0]], P->Y[sp4[1]], P->Y[sp4[2]], P->Y[sp4[3]], P->Y[INERT]};
            for (int j = 0; j < 6; j++) { vf[j] = af6[j]; sch[j] = j == 0 ? 2 : 3; }
            cgx[0] = g2x[0]; cgy[0] = g2y[0]; cgz[0] = g2z[0]; cgx[5] = g2x[1]; cgy[5] = g2y[1]; cgz[5] = g2z[1];
            for (int j = 0; j < n4; j++) { cgx[1 + j] = ggx[j]; cgy[1 + j] = ggy[j]; cgz[1 + j] = ggz[j]; }
            FFM_TRY(ffm_fv_multivariate_weights(m, 6, sch, 1.0, 0.0, 1.0, P->phi, vf, cgx, cgy, cgz, P->wMv));
        } else {
            double *gx = P->wN[1], *gy = P->wN[2], *gz = P->wN[3], *lim = P->wF[1];
            const double *fld[6] = {P->hs, P->Y[sp4[0]], P->Y[sp4[1]], P->Y[sp4[2]], P->Y[sp4[3]], P->Y[INERT]};
            const double *fb[6] = {hb, P->spB[0], P->spB[1], P->spB[2], P->spB[3], Nb};
            for (int j = 0; j < 6; j++) {
                FFM_TRY(ffm_fvc_grad(m, fld[j], fb[j], gx, gy, gz));
                FFM_TRY(HX(P, gx)); FFM_TRY(HX(P, gy)); FFM_TRY(HX(P, gz));
                FFM_TRY(ffm_fv_limited_limiter(m, j == 0 ? 2 : 3, 1.0, 0.0, 1.0, P->phi, fld[j], gx, gy, gz, lim, j == 0 ? 0 : 1));
            }
            FFM_TRY(ffm_fv_weights_from_limiter(m, P->phi, lim, P->wMv));
        }
    }
    if (P->fused) {
        // the four transported species share phi, rho and dEff: boundary values, gradients and matrices of all four in one
        // pass each, then the solves in the reference's order (nothing a later equation reads changes in an earlier solve)
        int sp[NSP - 1], ns = 0;
        for (int i = 0; i < NSP; i++) if (i != INERT) sp[ns++] = i;
        const double *vf[4], *vb[4], *vf0[4], *fq[4], *rq[4], *gq[4], *suq[4], *cgx[4], *cgy[4], *cgz[4];
        double *ggx[4], *ggy[4], *ggz[4];
        for (int j = 0; j < ns; j++) {
            const int i = sp[j]; const double nu = NU[i]; double *sj = P->suM[j];
            forN(P, N, [=] __device__(long c) { sj[c] = nu * wFuel[c]; });
            if (!P->mvSelection) FFM_TRY(ffm_bc_values(m, P->fS, P->refY[i], P->zeroB, P->Y[i], P->spB[j]));
            vf[j] = P->Y[i]; vb[j] = P->spB[j]; vf0[j] = P->Y0[i]; fq[j] = P->fS; rq[j] = P->refY[i]; gq[j] = P->zeroB; suq[j] = sj;
            ggx[j] = P->gM[j][0]; ggy[j] = P->gM[j][1]; ggz[j] = P->gM[j][2]; cgx[j] = ggx[j]; cgy[j] = ggy[j]; cgz[j] = ggz[j];
        }
        // with the common weights and one diffusivity the four species have the SAME off-diagonal coefficients (only diag and source
        // differ, through the patch conditions and the sources): written once, gathered into the sweeps' layout once
        double *uShared[4] = {P->spU[0], nullptr, nullptr, nullptr}, *lShared[4] = {P->spL[0], nullptr, nullptr, nullptr};
        if (P->mvSelection)
            FFM_TRY(ffm_fvm_scalar_transport_multi_w(m, ns, P->wMv, rdt, P->rho, P->rho0, P->phi, P->phib, af, afb, vf0, fq, rq, gq, suq, nullptr, nullptr,
                                                     nullptr, P->spD, uShared, lShared, P->spS));
        else {
        FFM_TRY(ffm_fvc_grad_multi(m, ns, vf, vb, ggx, ggy, ggz));
        for (int j = 0; j < ns; j++) { FFM_TRY(HX(P, ggx[j])); FFM_TRY(HX(P, ggy[j])); FFM_TRY(HX(P, ggz[j])); }
        FFM_TRY(ffm_fvm_scalar_transport_multi(m, ns, 3, 1.0, 0.0, 1.0, rdt, P->rho, P->rho0, P->phi, P->phib, af, afb, vf, cgx, cgy, cgz, vf0,
                                               fq, rq, gq, suq, nullptr, nullptr, nullptr, P->spD, P->spU, P->spL, P->spS));
        }
        if (P->mvSelection) {          // common weights, one diffusivity: the species' systems differ in diagonal and source only
            const char *nmq[4]; const double *dq[4], *sq[4]; double *pq[4];
            for (int j = 0; j < ns; j++) { nmq[j] = SPN[sp[j]]; dq[j] = P->spD[j]; sq[j] = P->spS[j]; pq[j] = P->Y[sp[j]]; }
            FFM_TRY(solve_named_multi(P, ns, nmq, 1e-8, dq, P->spU[0], P->spL[0], pq, sq));
        }
        for (int j = 0; j < ns; j++) {
            const int i = sp[j];
            if (!P->mvSelection) FFM_TRY(solve_named(P, SPN[i], FFM_PBICGSTAB, FFM_DILU, 1e-8, 0.0, P->spD[j], P->spU[j], P->spL[j], P->Y[i], P->spS[j]));
            FFM_TRY(HX(P, P->Y[i]));
            double *Yi = P->Y[i];
            forN(P, N, [=] __device__(long c) { const double v = fmax(Yi[c], 0.0); Yi[c] = v; Yt[c] += v; });
        }
    } else
    for (int i = 0; i < NSP; i++) {
        if (i == INERT) continue;
        const double nu = NU[i];
        forN(P, N, [=] __device__(long c) { su[c] = nu * wFuel[c]; });
        FFM_TRY(scalar_transport(P, SPN[i], 3, P->Y[i], P->Y0[i], P->fS, P->refY[i], af, afb, su, nullptr, 1e-8, nullptr, nullptr, P->mvSelection ? P->wMv : nullptr));
        double *Yi = P->Y[i];
        forN(P, N, [=] __device__(long c) { const double v = fmax(Yi[c], 0.0); Yi[c] = v; Yt[c] += v; });
    }
    {
        double *Yn = P->Y[INERT];
        forN(P, N, [=] __device__(long c) { Yn[c] = fmax(1.0 - Yt[c], 0.0); });
    }
    bool speciesOffDiagBound = P->fused && P->mvSelection;      // the species' off-diagonals are the matrix' bound (and gathered) ones
    if (P->radFreq > 0 && P->stepNo % P->radFreq == 0) { FFM_TRY(radiation_correct(P)); speciesOffDiagBound = false; }       // radiation->correct(), solver/YEEqn.H:80
    // EEqn: explicit LHS terms fvc::ddt(rho,K) + fvc::div(phi,K) - dpdt
    {
        FFM_TRY(U_boundary(P, Ub));     // U.correctBoundaryConditions() after the momentum solve
        double *Kb = P->wB[0], *kgx = P->wN[1], *kgy = P->wN[2], *kgz = P->wN[3], *wK = P->wF[3], *Kf = P->wF[4], *KfB = P->wB[5];
        const double *b0 = Ub[0], *b1 = Ub[1], *b2 = Ub[2];
        forN(P, B, [=] __device__(long k) { Kb[k] = 0.5 * ((b0[k] * b0[k] + b1[k] * b1[k]) + b2[k] * b2[k]); });
        FFM_TRY(ffm_fvc_grad(m, P->K, Kb, kgx, kgy, kgz));
        FFM_TRY(HX(P, kgx)); FFM_TRY(HX(P, kgy)); FFM_TRY(HX(P, kgz));
        FFM_TRY(ffm_fv_limited_weights(m, 2, 1.0, 0.0, 1.0, P->phi, P->K, kgx, kgy, kgz, wK));
        FFM_TRY(ffm_fvc_interpolate(m, wK, P->K, Kf));
        const double *phi = P->phi, *phib = P->phib;
        forN(P, nNat, [=] __device__(long e) { Kf[e] = phi[e] * Kf[e]; });
        forN(P, B, [=] __device__(long k) { KfB[k] = phib[k] * Kb[k]; });
        double *divK = P->wN[4];
        FFM_TRY(ffm_fvc_surface_integrate(m, Kf, KfB, divK));
        double *ddtK = P->wN[0], *ndpdt = P->wN[5]; const double *rho = P->rho, *rho0 = P->rho0, *K = P->K, *K0 = P->K0, *dpdt = P->dpdt;
        forN(P, N, [=] __device__(long c) { ddtK[c] = rdt * (rho[c] * K[c] - rho0[c] * K0[c]); ndpdt[c] = -dpdt[c]; });
        const double *expl[3] = {ddtK, divK, ndpdt};       // fvc::ddt(rho,K) + fvc::div(phi,K) + (-dpdt), solver/YEEqn.H:89-101
        // + radiation->Sh(thermo, he) = Ru - fvm::Sp(4 Rp T^3/Cpv, he) - Rp T^3 (T - 4 he/Cpv), Rp = 4 a sigma, Ru = a G - E
        // (radiationModel.C:229-244, fvDOM.C Rp / Ru), E of the current Qdot
        const double *shSu = nullptr, *shSp = nullptr;
        if (P->radCoupled && P->radHaveG) {
            double frac = 0.0;
            FFM_TRY(plume_rad_fraction(P, &frac));
            const double Rp = 4.0 * P->radA * SIGMA_SB, KA = P->radA;
            double *su2 = P->radShSu, *sp2 = P->radShSp; const double *G = P->G, *T = P->T, *hh = P->hs, *Qd = Qdot;
            forN(P, N, [=] __device__(long c) {
                const double t = T[c], T3 = t * t * t;
                const double Ru = KA * G[c] - frac * Qd[c];
                sp2[c] = 4.0 * Rp * T3 / CP;
                su2[c] = Ru - Rp * T3 * (t - 4.0 * hh[c] / CP);
            });
            shSu = su2; shSp = sp2;
        }
        if (P->fused) {
            const double *vf[1] = {P->hs}, *vb[1] = {P->spB[0]}, *vf0[1] = {P->hs0}, *fq[1] = {P->fH}, *rq[1] = {P->refH}, *gq[1] = {P->zeroB}, *suq[1] = {Qdot};
            double *ggx[1] = {P->gM[0][0]}, *ggy[1] = {P->gM[0][1]}, *ggz[1] = {P->gM[0][2]};
            const double *cgx[1] = {ggx[0]}, *cgy[1] = {ggy[0]}, *cgz[1] = {ggz[0]}, *su2q[1] = {shSu}, *spq[1] = {shSp};
            double *dd[1] = {P->dWork}, *uu[1] = {P->upper}, *ll[1] = {P->lower}, *ss[1] = {P->sWork};
            // h is convected with the same weights and diffuses with the same alphaEff (Le = 1: af) as the species: where their
            // off-diagonals are still the bound ones (no ray solve in between) the enthalpy matrix shares them as well
            const bool shareH = speciesOffDiagBound;
            if (shareH) { uu[0] = nullptr; ll[0] = nullptr; }
            if (P->mvSelection)
                FFM_TRY(ffm_fvm_scalar_transport_multi_w(m, 1, P->wMv, rdt, P->rho, P->rho0, P->phi, P->phib, af, afb, vf0, fq, rq, gq, suq, su2q, spq, expl,
                                                         dd, uu, ll, ss));
            else {
            FFM_TRY(ffm_bc_values(m, P->fH, P->refH, P->zeroB, P->hs, P->spB[0]));
            FFM_TRY(ffm_fvc_grad_multi(m, 1, vf, vb, ggx, ggy, ggz));
            FFM_TRY(HX(P, ggx[0])); FFM_TRY(HX(P, ggy[0])); FFM_TRY(HX(P, ggz[0]));
            FFM_TRY(ffm_fvm_scalar_transport_multi(m, 1, 2, 1.0, 0.0, 1.0, rdt, P->rho, P->rho0, P->phi, P->phib, af, afb, vf, cgx, cgy, cgz, vf0,
                                                   fq, rq, gq, suq, su2q, spq, expl, dd, uu, ll, ss));
            }
            if (shareH) FFM_TRY(solve_named(P, "h", FFM_PBICGSTAB, FFM_DILU, 1e-8, 0.0, P->dWork, P->spU[0], P->spL[0], P->hs, P->sWork, true));
            else FFM_TRY(solve_named(P, "h", FFM_PBICGSTAB, FFM_DILU, 1e-8, 0.0, P->dWork, P->upper, P->lower, P->hs, P->sWork));
            FFM_TRY(HX(P, P->hs));
        } else
        FFM_TRY(scalar_transport(P, "h", 2, P->hs, P->hs0, P->fH, P->refH, af, afb, Qdot, expl, 1e-8, shSu, shSp, P->mvSelection ? P->wMv : nullptr));
    }
    standin_thermo(P);
    { const char *st = getenv("FFM_PLUME_STOP"); if (st && atoi(st) == 2) { PL_HIP(hipStreamSynchronize(P->ctx->stream)); return FFM_OK; } }
    // ---------------- pEqn.H x 2
    FFM_TRY(p_corrector(P, false));
    { const char *st = getenv("FFM_PLUME_STOP"); if (st && atoi(st) == 3) { PL_HIP(hipStreamSynchronize(P->ctx->stream)); return FFM_OK; } }
    FFM_TRY(p_corrector(P, true));
    mul(P, P->rho, P->psi, P->p, N);
    P->time += P->dt; P->stepNo++;
    PL_HIP(hipStreamSynchronize(P->ctx->stream));
    return FFM_OK;
}

extern "C" int ffm_plume_create(ffm_ctx *ctx, int nx, int ny, int nz, double h, double dt, ffm_plume **out)
{
    const int lo[3] = {0, 0, 0}, hi[3] = {nx, ny, nz}, nbr[6] = {-1, -1, -1, -1, -1, -1};
    return ffm_plume_create_block(ctx, nx, ny, nz, lo, hi, nbr, h, dt, out);
}

// One rank's block [lo,hi) of the global (gx,gy,gz) box.  nbrRank[6] = rank across the -x,+x,-y,+y,-z,+z side of the
// block, or -1 where that side is a physical boundary of the box.  Ghost layers carry the neighbour ranks' cells.
extern "C" int ffm_plume_create_block(ffm_ctx *ctx, int gx, int gy, int gz, const int *lo, const int *hi, const int *nbrRank,
                                      double h, double dt, ffm_plume **out)
{
    if (!ctx || !out || !lo || !hi || !nbrRank) return FFM_ERR_ARG;
    const int nx = hi[0] - lo[0], ny = hi[1] - lo[1], nz = hi[2] - lo[2];
    if (nx < 2 || ny < 2 || nz < 2) { ffm_set_error("plume block must be at least 2 cells wide in every direction"); return FFM_ERR_ARG; }
    for (int d = 0; d < 3; d++) {
        const int G = d == 0 ? gx : d == 1 ? gy : gz;
        if ((lo[d] == 0) != (nbrRank[2 * d] < 0) || (hi[d] == G) != (nbrRank[2 * d + 1] < 0)) { ffm_set_error("plume block: neighbour ranks inconsistent with the block position"); return FFM_ERR_ARG; }
    }
    PL_HIP(hipSetDevice(ctx->device));
    FfmStageTimer tmAll_("plume_create: total");
    ffm_plume *P = new ffm_plume();
    P->ctx = ctx; P->nx = nx; P->ny = ny; P->nz = nz; P->h = h; P->dt = dt; P->rdt = 1.0 / dt;
    P->tight = getenv("FFM_PLUME_TIGHT") != nullptr;      // tests only: see ffm_plume_set_tight
    if (const char *e = getenv("FFM_PLUME_SOLVERS")) P->stecklerSolvers = e[0] == 's';     // "steckler": see ffm_plume_set_solvers
    const long nOwn = (long)nx * ny * nz;
    // ---- ghost layers, one per coupled side, in side order -x,+x,-y,+y,-z,+z; inside a layer in natural order
    const int cnt[6] = {ny * nz, ny * nz, nx * nz, nx * nz, nx * ny, nx * ny};
    int gOff[7]; gOff[0] = 0;
    for (int s6 = 0; s6 < 6; s6++) gOff[s6 + 1] = gOff[s6] + (nbrRank[s6] >= 0 ? cnt[s6] : 0);
    const long nGhost = gOff[6], N = nOwn + nGhost;
    P->N = (int)N; P->nOwn = (int)nOwn;
    auto cellOf = [&](int i, int j, int k) { return i + nx * (j + ny * k); };
    auto ghostOf = [&](int side, int i, int j, int k) -> int {     // ghost across `side` of owned cell (i,j,k)
        switch (side >> 1) {
        case 0: return (int)nOwn + gOff[side] + j + ny * k;
        case 1: return (int)nOwn + gOff[side] + i + nx * k;
        default: return (int)nOwn + gOff[side] + i + nx * j;
        }
    };
    FfmStageTimer *tmS_ = new FfmStageTimer("plume_create: natural LDU");
    // ---- LDU in local natural order + cut faces owned by the owned cell (ghost index > every owned index)
    std::vector<int> l, u; std::vector<signed char> fd, fsgn;
    l.reserve(3 * nOwn + nGhost); u.reserve(3 * nOwn + nGhost); fd.reserve(3 * nOwn + nGhost); fsgn.reserve(3 * nOwn + nGhost);
    for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) {
        const int c = cellOf(i, j, k);
        if (i < nx - 1) { l.push_back(c); u.push_back(c + 1); fd.push_back(0); fsgn.push_back(1); }
        if (j < ny - 1) { l.push_back(c); u.push_back(c + nx); fd.push_back(1); fsgn.push_back(1); }
        if (k < nz - 1) { l.push_back(c); u.push_back(c + nx * ny); fd.push_back(2); fsgn.push_back(1); }
        const int at[6] = {i == 0, i == nx - 1, j == 0, j == ny - 1, k == 0, k == nz - 1};
        for (int s6 = 0; s6 < 6; s6++) if (at[s6] && nbrRank[s6] >= 0) {    // ascending ghost index = ascending side
            l.push_back(c); u.push_back(ghostOf(s6, i, j, k)); fd.push_back((signed char)(s6 >> 1)); fsgn.push_back((s6 & 1) ? 1 : -1);
        }
    }
    const int F = (int)l.size(); P->F = F;
    delete tmS_; tmS_ = nullptr;
    // ---- renumber once to the library's cell order (no permutation pass ever after)
    // group hint for the tiled sweeps: 2-D tiles of x-columns, FFM_TILE x FFM_TILE cells in (y,z) (ignored by the level mode)
    int tileEdge = 16;
    if (const char *e = getenv("FFM_TILE")) tileEdge = std::max(1, atoi(e));
    std::vector<int> hint(nOwn);
    for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) hint[cellOf(i, j, k)] = ffm_tile_label(j / tileEdge, k / tileEdge);
    std::vector<int> c2(N), f2(F);
    { FfmStageTimer tm_("plume_create: renumber_hint"); FFM_TRY(ffm_renumber_hint((int)nOwn, (int)nGhost, F, l.data(), u.data(), hint.data(), c2.data(), f2.data())); }
    P->newToOld = c2; P->faceNewToOld = f2;
    tmS_ = new FfmStageTimer("plume_create: renumbered addressing");
    std::vector<int> oldToNew(N);
    for (long c = 0; c < N; c++) oldToNew[c2[c]] = (int)c;
    std::vector<int> l2(F), u2(F); std::vector<signed char> fd2(F), sg2(F);
    { const int *o2n = oldToNew.data(), *lp = l.data(), *up_ = u.data(), *f2p = f2.data(); const signed char *fdp = fd.data(), *sgp = fsgn.data();
      int *l2p = l2.data(), *u2p = u2.data(); signed char *fd2p = fd2.data(), *sg2p = sg2.data();
      ffm_parallel_for(F, [=](long a, long b) { for (long f = a; f < b; f++) { l2p[f] = o2n[lp[f2p[f]]]; u2p[f] = o2n[up_[f2p[f]]]; fd2p[f] = fdp[f2p[f]]; sg2p[f] = sgp[f2p[f]]; } }); }
    delete tmS_; tmS_ = nullptr;
    {
        std::vector<int> hintNew(nOwn);
        for (long c = 0; c < nOwn; c++) hintNew[c] = hint[c2[c]];
        FfmStageTimer tm_("plume_create: ldu_create");
        FFM_TRY(ffm_ldu_create_hint(ctx, (int)nOwn, (int)nGhost, F, l2.data(), u2.data(), hintNew.data(), &P->A));
    }
    if (!P->A->identity) { ffm_set_error("plume: renumbered mesh is not native"); return FFM_ERR_ADDR; }
    if (nGhost == 0) { P->hL2 = l2; P->hU2 = u2; P->hOldToNew = oldToNew; P->hFd2 = fd2; }      // for the direction-ordered ray solves
    FFM_TRY(ffm_ldu_set_global_cells(P->A, (long)gx * gy * gz));
    P->nNat = P->A->upTotal;
    tmS_ = new FfmStageTimer("plume_create: geometry + patches");
    // ---- geometry (global coordinates)
    std::vector<double> V(N, h * h * h), C(3 * N), Sf(3 * (size_t)F, 0.0), magSf(F, h * h), wgt(F, 0.5), del(F, 1.0 / h), Cfy(F);
    {
        double *Cp = C.data(), *Sfp = Sf.data(), *Cfyp = Cfy.data(); const int *c2p = c2.data(), *l2p = l2.data(); const signed char *fd2p = fd2.data(), *sg2p = sg2.data();
        const int lo0 = lo[0], lo1 = lo[1], lo2 = lo[2]; int gO[7]; for (int q = 0; q < 7; q++) gO[q] = gOff[q];
        ffm_parallel_for(N, [=](long a, long b) { for (long cn = a; cn < b; cn++) {
            const int co = c2p[cn];
            int i, j, k;
            if (co < nOwn) { i = co % nx; j = (co / nx) % ny; k = co / (nx * ny); }
            else {
                int side = 0; while (co - nOwn >= gO[side + 1]) side++;
                const int r = co - (int)nOwn - gO[side];
                switch (side >> 1) {
                case 0: j = r % ny; k = r / ny; i = (side & 1) ? nx : -1; break;
                case 1: i = r % nx; k = r / nx; j = (side & 1) ? ny : -1; break;
                default: i = r % nx; j = r / nx; k = (side & 1) ? nz : -1; break;
                }
            }
            Cp[cn] = (lo0 + i + 0.5) * h; Cp[N + cn] = (lo1 + j + 0.5) * h; Cp[2 * N + cn] = (lo2 + k + 0.5) * h;
        } });
        ffm_parallel_for(F, [=](long a, long b) { for (long f = a; f < b; f++) { Sfp[(size_t)fd2p[f] * F + f] = sg2p[f] * h * h; Cfyp[f] = Cp[N + l2p[f]] + (fd2p[f] == 1 ? sg2p[f] * 0.5 * h : 0.0); } });
    }
    // ---- ghost exchange plan: side s sends the owned layer adjacent to it, receives the ghost layer across it
    {
        std::vector<int> ranks, sc, rc, cells;
        for (int s6 = 0; s6 < 6; s6++) if (nbrRank[s6] >= 0) {
            ranks.push_back(nbrRank[s6]); sc.push_back(cnt[s6]); rc.push_back(cnt[s6]);
            const int d = s6 >> 1, pos = (s6 & 1) ? (d == 0 ? nx : d == 1 ? ny : nz) - 1 : 0;
            if (d == 0) for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) cells.push_back(oldToNew[cellOf(pos, j, k)]);
            else if (d == 1) for (int k = 0; k < nz; k++) for (int i = 0; i < nx; i++) cells.push_back(oldToNew[cellOf(i, pos, k)]);
            else for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) cells.push_back(oldToNew[cellOf(i, j, pos)]);
        }
        if (!ranks.empty()) FFM_TRY(ffm_ldu_set_ghost_exchange(P->A, (int)ranks.size(), ranks.data(), sc.data(), cells.data(), rc.data()));
    }
    // ---- patches on the physical sides of the block: inlet, floor, top, sides(xmin,xmax,zmin,zmax)
    const double Lx = gx * h, Lz = gz * h, hwx = std::min(0.5, Lx / 4), hwz = std::min(0.5, Lz / 4);
    std::vector<int> pc[4]; std::vector<double> pS[4][3];
    auto addFace = [&](int patch, int cOld, double sx, double sy, double sz) {
        pc[patch].push_back(oldToNew[cOld]); pS[patch][0].push_back(sx); pS[patch][1].push_back(sy); pS[patch][2].push_back(sz);
    };
    if (nbrRank[2] < 0) for (int k = 0; k < nz; k++) for (int i = 0; i < nx; i++) {          // ymin in natural cell order
        const bool in = std::fabs((lo[0] + i + 0.5) * h - Lx / 2) < hwx && std::fabs((lo[2] + k + 0.5) * h - Lz / 2) < hwz;
        addFace(in ? P_INLET : P_FLOOR, cellOf(i, 0, k), 0, -h * h, 0);
    }
    if (nbrRank[3] < 0) for (int k = 0; k < nz; k++) for (int i = 0; i < nx; i++) addFace(P_TOP, cellOf(i, ny - 1, k), 0, h * h, 0);
    if (nbrRank[0] < 0) for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) addFace(P_SIDES, cellOf(0, j, k), -h * h, 0, 0);
    if (nbrRank[1] < 0) for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) addFace(P_SIDES, cellOf(nx - 1, j, k), h * h, 0, 0);
    if (nbrRank[4] < 0) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) addFace(P_SIDES, cellOf(i, j, 0), 0, 0, -h * h);
    if (nbrRank[5] < 0) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) addFace(P_SIDES, cellOf(i, j, nz - 1), 0, 0, h * h);
    int sizes[4]; const int *fcs[4]; const double *pSf[4]; const double *pDel[4];
    std::vector<double> pSflat[4], pD[4];
    int Btot = 0;
    for (int p = 0; p < 4; p++) {
        sizes[p] = (int)pc[p].size(); fcs[p] = pc[p].data(); Btot += sizes[p];
        for (int d = 0; d < 3; d++) pSflat[p].insert(pSflat[p].end(), pS[p][d].begin(), pS[p][d].end());
        pD[p].assign(sizes[p], 2.0 / h); pSf[p] = pSflat[p].data(); pDel[p] = pD[p].data();
    }
    P->B = Btot;
    delete tmS_; tmS_ = nullptr;
    { FfmStageTimer tm_("plume_create: mesh_create"); FFM_TRY(ffm_mesh_create(P->A, V.data(), C.data(), Sf.data(), magSf.data(), wgt.data(), del.data(), 4, sizes, fcs, pSf, pDel, &P->mesh)); }
    tmS_ = new FfmStageTimer("plume_create: face centres");
    {
        // face centres (LUST correction): owner's centre + half a cell towards the neighbour
        std::vector<double> Cf(3 * (size_t)F);
        { double *Cfp = Cf.data(); const double *Cp = C.data(); const int *l2p = l2.data(); const signed char *fd2p = fd2.data(), *sg2p = sg2.data();
          ffm_parallel_for(F, [=](long a, long b) { for (long f = a; f < b; f++) for (int d = 0; d < 3; d++) Cfp[(size_t)d * F + f] = Cp[(size_t)d * N + l2p[f]] + (fd2p[f] == d ? sg2p[f] * 0.5 * h : 0.0); }); }
        FFM_TRY(ffm_mesh_set_face_centres(P->mesh, Cf.data()));
    }
    delete tmS_; tmS_ = new FfmStageTimer("plume_create: fields + tables");
    const int ny_glob = gy;
    const int B = Btot; const long nNat = P->nNat;
    // ---- fields
    auto NN = [&]() { return dalloc(P, N); };
    for (int i = 0; i < NSP; i++) { P->Y[i] = dfill(P, N, Y_AMB[i]); P->Y0[i] = NN(); }
    P->T = dfill(P, N, TREF); P->hs = NN(); P->hs0 = NN();
    for (int c = 0; c < 3; c++) { P->U[c] = NN(); P->U0[c] = NN(); }
    P->p = dfill(P, N, PREF); P->p0 = NN(); P->p_rgh = NN(); P->p_rgh0 = NN(); P->psi = NN(); P->psi0 = NN();
    P->rho = NN(); P->rho0 = NN(); P->K = NN(); P->K0 = NN(); P->dpdt = NN(); P->ph_rgh = NN();
    P->phi = dalloc(P, nNat); P->phi0 = dalloc(P, nNat); P->phib = dalloc(P, B); P->phib0 = dalloc(P, B); P->ph_rgh_b = dalloc(P, B);
    {
        const double ghRef = -9.81 * (ny_glob * h);
        std::vector<double> gh(N), ghf(std::max<long>(nNat, 1), 0.0);
        for (long c = 0; c < N; c++) gh[c] = -9.81 * C[N + c] - ghRef;
        for (int f = 0; f < F; f++) ghf[P->A->h_callerToNative[f]] = -9.81 * Cfy[f] - ghRef;
        P->gh = dupload(P, gh); P->ghf = dupload(P, ghf);
    }
    // ---- boundary-condition templates
    std::vector<double> kind(B), fsU[3], rU[3], fsS(B), fsH(B), rY[NSP], rH(B);
    for (int c = 0; c < 3; c++) { fsU[c].assign(B, 0.0); rU[c].assign(B, 0.0); }
    for (int i = 0; i < NSP; i++) rY[i].assign(B, 0.0);
    int k0 = 0;
    for (int p = 0; p < 4; p++) for (int q = 0; q < sizes[p]; q++, k0++) {
        kind[k0] = p;
        for (int c = 0; c < 3; c++) {
            if (p == P_INLET) { fsU[c][k0] = 1.0; rU[c][k0] = (c == 1) ? U_IN : 0.0; }
            else if (p == P_FLOOR) fsU[c][k0] = 1.0;
            else fsU[c][k0] = (pS[p][c][q] != 0.0) ? 0.0 : -1.0;      // normal: zeroGradient; tangential: inletOutlet(0)
        }
        fsH[k0] = (p == P_INLET || p == P_FLOOR) ? 1.0 : -1.0;
        if (p == P_INLET) { fsS[k0] = 1.0; rH[k0] = CP * (T_IN - TREF); for (int i = 0; i < NSP; i++) rY[i][k0] = Y_IN[i]; }
        else if (p == P_FLOOR) { fsS[k0] = 0.0; }                     // species zeroGradient; h handled below
        else { fsS[k0] = -1.0; rH[k0] = 0.0; for (int i = 0; i < NSP; i++) rY[i][k0] = Y_AMB[i]; }
    }
    P->kind_d = dupload(P, kind);
    for (int c = 0; c < 3; c++) { P->fStaticU[c] = dupload(P, fsU[c]); P->refU[c] = dupload(P, rU[c]); P->fU[c] = dalloc(P, B); }
    P->fStaticS = dupload(P, fsS); P->fS = dalloc(P, B); P->refH = dupload(P, rH);
    P->fStaticH = dupload(P, fsH); P->fH = dalloc(P, B);
    for (int i = 0; i < NSP; i++) P->refY[i] = dupload(P, rY[i]);
    P->fP = dalloc(P, B); P->refP = dalloc(P, B); P->gradP = dalloc(P, B); P->zeroB = dalloc(P, B);
    P->oneB = dupload(P, std::vector<double>(std::max(B, 1), 1.0));
    // ---- matrix + work
    P->diag = NN(); P->upper = dalloc(P, nNat); P->lower = dalloc(P, nNat); P->dWork = NN(); P->sWork = NN();
    P->Udiag = NN(); P->Uupper = dalloc(P, nNat); P->Ulower = dalloc(P, nNat);
    for (int c = 0; c < 3; c++) { P->src[c] = NN(); P->ic[c] = dalloc(P, B); P->bc[c] = dalloc(P, B); P->Usrc[c] = NN(); P->Uic[c] = dalloc(P, B); P->Ubc[c] = dalloc(P, B); }
    for (auto &w : P->wN) w = NN();
    for (auto &w : P->wF) w = dalloc(P, nNat);
    P->ddtCorrF = dalloc(P, nNat);
    for (auto &w : P->wB) w = dalloc(P, B);
    P->fused = getenv("FFM_PLUME_UNFUSED") == nullptr;
    P->mvSelection = getenv("FFM_PLUME_INDEPENDENT_LIMITERS") == nullptr;
    if (P->mvSelection) { P->wMv = dalloc(P, nNat); if (P->fused) for (int j = 0; j < 2; j++) for (int d = 0; d < 3; d++) P->mvG[j][d] = NN(); }
    for (int j = 0; j < 4; j++) {
        for (int d = 0; d < 3; d++) P->gM[j][d] = P->fused ? NN() : nullptr;
        P->spD[j] = P->fused ? NN() : nullptr; P->spS[j] = P->fused ? NN() : nullptr; P->suM[j] = P->fused ? NN() : nullptr;
        const bool ownOffDiag = P->fused && (j == 0 || !P->mvSelection);       // common limiter: the species share one pair of off-diagonal arrays
        P->spU[j] = ownOffDiag ? dalloc(P, nNat) : nullptr; P->spL[j] = ownOffDiag ? dalloc(P, nNat) : nullptr; P->spB[j] = (P->fused || P->mvSelection) ? dalloc(P, B) : nullptr;
    }
    for (double *p : P->pool) if (!p) { ffm_set_error("plume: out of device memory"); return FFM_ERR_HIP; }
    PL_HIP(hipDeviceSynchronize());
    delete tmS_; tmS_ = new FfmStageTimer("plume_create: hydrostatic init");
    // ---- initial state: quiescent ambient, then hydrostatic initialisation
    standin_thermo(P);
    mul(P, P->rho, P->psi, P->p, N);
    FFM_TRY(hydrostatic_init(P));
    PL_HIP(hipStreamSynchronize(ctx->stream));
    delete tmS_; tmS_ = nullptr;
    if (const char *e = getenv("FFM_PLUME_RADIATION")) { if (atoi(e) > 0) FFM_TRY(ffm_plume_set_radiation(P, atoi(e), 2, 4, nullptr, nullptr)); }   // solverFreq
    *out = P;
    return FFM_OK;
}

extern "C" int ffm_plume_destroy(ffm_plume *P)
{
    if (!P) return FFM_OK;
    hipStreamSynchronize(P->ctx->stream);
    for (double *p : P->pool) hipFree(p);
    for (int a = 0; a < 3; a++) { hipFree(P->radCm[a]); hipFree(P->radFm[a]); }
    ffm_mesh_destroy(P->mesh); ffm_ldu_destroy(P->A);
    delete P;
    return FFM_OK;
}

extern "C" int ffm_plume_set_tight(ffm_plume *P, int on) { if (!P) return FFM_ERR_ARG; P->tight = on != 0; return FFM_OK; }
// Switch the fvDOM stand-in on: every `solverFreq` steps (cases/steckler/constant/radiationProperties:32-40: solverFreq 100,
// nPhi 2, nTheta 4 -> 32 rays) the step solves one upwind transport equation per ray before the enthalpy equation.
// dAve[3*nRay] / omega[nRay] may be given by the caller (the shim passes fvDOM's own); null -> built here from nPhi, nTheta.
extern "C" int ffm_plume_set_radiation(ffm_plume *P, int solverFreq, int nPhi, int nTheta, const double *dAve, const double *omega)
{
    if (!P || solverFreq < 0 || nPhi < 1 || nTheta < 1 || ((dAve == nullptr) != (omega == nullptr))) return FFM_ERR_ARG;
    PL_HIP(hipSetDevice(P->ctx->device));
    const int nRay = 4 * nPhi * nTheta;
    P->rayD.assign(3 * (size_t)nRay, 0.0); P->rayOmega.assign(nRay, 0.0);
    if (dAve) { std::copy(dAve, dAve + 3 * nRay, P->rayD.begin()); std::copy(omega, omega + nRay, P->rayOmega.begin()); }
    else {
        const double dPhi = M_PI / (2.0 * nPhi), dTheta = M_PI / nTheta; int i = 0;
        for (int n = 1; n <= nTheta; n++) for (int mm = 1; mm <= 4 * nPhi; mm++, i++) {
            const double theta = (2.0 * n - 1.0) * dTheta / 2.0, phi = (2.0 * mm - 1.0) * dPhi / 2.0;
            const double a = sin(0.5 * dPhi) * (dTheta - cos(2.0 * theta) * sin(dTheta));
            P->rayOmega[i] = 2.0 * sin(theta) * sin(dTheta / 2.0) * dPhi;
            P->rayD[3 * i] = sin(phi) * a; P->rayD[3 * i + 1] = cos(phi) * a; P->rayD[3 * i + 2] = 0.5 * dPhi * sin(2.0 * theta) * sin(dTheta);
        }
    }
    if ((int)P->I.size() < nRay) {
        for (int i = (int)P->I.size(); i < nRay; i++) { double *p = dalloc(P, P->N); if (!p) return FFM_ERR_HIP; P->I.push_back(p); }
    }
    if (!P->G) {
        P->G = dalloc(P, P->N); P->radSrc = dalloc(P, P->N); P->radJ = dalloc(P, P->nNat); P->radW = dalloc(P, P->nNat);
        P->radJb = dalloc(P, P->B); P->radF = dalloc(P, P->B); P->radRef = dalloc(P, P->B);
        if (!P->G || !P->radSrc || !P->radJ || !P->radW || !P->radJb || !P->radF || !P->radRef) return FFM_ERR_HIP;
    }
    // ---- direction-ordered solves: one block without ghost layers (a decomposed block keeps the iterative solve: the
    // exact sweep would have to cross rank boundaries)
    if (!P->hL2.empty() && !P->radOrdered && !getenv("FFM_RAD_ITERATIVE")) {
        const int N = P->N, F = P->F, nx = P->nx, ny = P->ny, nz = P->nz; const long nNat = P->nNat;
        std::vector<int> ownerStart(N + 1, 0);
        for (int f = 0; f < F; f++) ownerStart[P->hL2[f] + 1]++;
        for (int c = 0; c < N; c++) ownerStart[c + 1] += ownerStart[c];            // faces are sorted by owner (upper-triangular order)
        const std::vector<int> &c2n = P->A->h_callerToNative;
        for (int a = 0; a < 3; a++) {
            std::vector<int> cm(N), fm(std::max<long>(nNat, 1));
            for (long e = 0; e < nNat; e++) fm[e] = (int)e;                        // padding entries map to themselves
            for (int c = 0; c < N; c++) {
                const int o = P->newToOld[c]; int i = o % nx, j = (o / nx) % ny, k = o / (nx * ny);
                if (a == 0) i = nx - 1 - i; else if (a == 1) j = ny - 1 - j; else k = nz - 1 - k;
                cm[c] = P->hOldToNew[i + nx * (j + ny * k)];
            }
            for (int f = 0; f < F; f++) {
                int o = cm[P->hL2[f]], n = cm[P->hU2[f]]; bool swap = false;
                if (o > n) { std::swap(o, n); swap = true; }
                int fp = -1;
                for (int g = ownerStart[o]; g < ownerStart[o + 1]; g++) if (P->hU2[g] == n) { fp = g; break; }
                if (fp < 0) { ffm_set_error("plume radiation: the flipped image of a face is not a face"); return FFM_ERR_ADDR; }
                fm[c2n[f]] = swap ? ~c2n[fp] : c2n[fp];
            }
            PL_HIP(hipMalloc((void **)&P->radCm[a], sizeof(int) * N)); PL_HIP(hipMalloc((void **)&P->radFm[a], sizeof(int) * std::max<long>(nNat, 1)));
            FFM_TRY(ffm_h2d(P->ctx, P->radCm[a], cm.data(), sizeof(int) * N));
            FFM_TRY(ffm_h2d(P->ctx, P->radFm[a], fm.data(), sizeof(int) * std::max<long>(nNat, 1)));
        }
        P->radDB = dalloc(P, N); P->radSB = dalloc(P, N); P->radPsiB = dalloc(P, N); P->radUB = dalloc(P, nNat); P->radLB = dalloc(P, nNat);
        if (!P->radDB || !P->radSB || !P->radPsiB || !P->radUB || !P->radLB) return FFM_ERR_HIP;
        P->radOrdered = true;
    }
    P->radFreq = solverFreq;
    return FFM_OK;
}

// the reference's absorption / emission model + radiation->Sh coupling (see include/ffm.h)
extern "C" int ffm_plume_set_radiation_model(ffm_plume *P, double absorption, double Ehrr1, double Ehrr2)
{
    if (!P || absorption < 0 || Ehrr1 < 0 || Ehrr2 < 0) return FFM_ERR_ARG;
    PL_HIP(hipSetDevice(P->ctx->device));
    if (!P->radE) { P->radE = dalloc(P, P->N); P->radShSu = dalloc(P, P->N); P->radShSp = dalloc(P, P->N); }
    if (!P->radE || !P->radShSu || !P->radShSp) return FFM_ERR_HIP;
    P->radA = absorption; P->Ehrr1 = Ehrr1; P->Ehrr2 = Ehrr2; P->radCoupled = true;
    return FFM_OK;
}
// Start state and boundary values other than the quiescent ambient / pure-fuel inflow (tests: a state in which no transported
// field is uniform, tests/test_plume_gpu.py).  Y[5], h: cell fields in natural blockMesh order of the (single) block; Yamb / Yin:
// the inletOutlet and inlet values of the species, hAmb the inletOutlet value of h.  Only before the first step: the hydrostatic
// initialisation (solver/phrghEqn.H) is redone from the new state.
extern "C" int ffm_plume_set_initial_state(ffm_plume *P, const double *const *Y, const double *h, const double *Yamb, const double *Yin, double hAmb)
{
    if (!P || !Y || !h || !Yamb || !Yin) return FFM_ERR_ARG;
    if (P->stepNo != 0 || P->N != P->nOwn) { ffm_set_error("plume: the start state can be set on a single block before the first step only"); return FFM_ERR_ARG; }
    PL_HIP(hipSetDevice(P->ctx->device));
    const int N = P->N, B = P->B;
    std::vector<double> v(N);
    auto up = [&](double *dst, const double *nat) -> int {
        for (int c = 0; c < N; c++) v[c] = nat[P->newToOld[c]];
        return ffm_memcpy_h2d(P->ctx, dst, v.data(), sizeof(double) * N);
    };
    for (int i = 0; i < NSP; i++) FFM_TRY(up(P->Y[i], Y[i]));
    FFM_TRY(up(P->hs, h));
    P->hAmb = hAmb;
    const double *kind = P->kind_d; double *rH = P->refH;
    for (int i = 0; i < NSP; i++) {
        double *r = P->refY[i]; const double a = Yamb[i], b = Yin[i];
        forN(P, B, [=] __device__(long k) { r[k] = kind[k] < 0.5 ? b : kind[k] > 1.5 ? a : 0.0; });
    }
    forN(P, B, [=] __device__(long k) { rH[k] = kind[k] < 0.5 ? CP * (T_IN - TREF) : kind[k] > 1.5 ? hAmb : 0.0; });
    double *p = P->p, *ph = P->ph_rgh;
    forN(P, N, [=] __device__(long c) { p[c] = PREF; ph[c] = 0.0; });
    P->log.clear();
    standin_thermo(P);
    mul(P, P->rho, P->psi, P->p, N);
    FFM_TRY(hydrostatic_init(P));
    PL_HIP(hipStreamSynchronize(P->ctx->stream));
    return FFM_OK;
}

// Tests (the deciding test of the multi-step parity question): the NEXT ffm_plume_step convects the species and h with these face
// weights instead of evaluating the multivariateSelection limiter; w[F] in natural blockMesh face order of the (single) block.
extern "C" int ffm_plume_override_mv_weights(ffm_plume *P, const double *w)
{
    if (!P || !w) return FFM_ERR_ARG;
    if (!P->mvSelection || P->N != P->nOwn) { ffm_set_error("plume: weights can be handed in on a single block with the common limiter only"); return FFM_ERR_ARG; }
    PL_HIP(hipSetDevice(P->ctx->device));
    std::vector<double> v(std::max<long>(P->nNat, 1), 0.0);
    const std::vector<int> &c2n = P->A->h_callerToNative;
    for (int f = 0; f < P->F; f++) v[c2n[f]] = w[P->faceNewToOld[f]];
    FFM_TRY(ffm_memcpy_h2d(P->ctx, P->wMv, v.data(), sizeof(double) * P->nNat));
    P->mvOverride = true;
    return FFM_OK;
}
extern "C" int ffm_plume_set_solvers(ffm_plume *P, int stecklerSelection) { if (!P) return FFM_ERR_ARG; P->stecklerSolvers = stecklerSelection != 0; return FFM_OK; }
extern "C" int ffm_plume_ncells(const ffm_plume *P) { return P ? P->nOwn : FFM_ERR_ARG; }
extern "C" int ffm_plume_nfaces(const ffm_plume *P) { return P ? P->F : FFM_ERR_ARG; }

// copy a cell field to the host in NATURAL blockMesh cell order; name in rho,p,p_rgh,T,h,K,Ux,Uy,Uz,psi,<specie>
extern "C" int ffm_plume_get_field(ffm_plume *P, const char *name, double *out)
{
    if (!P || !name || !out) return FFM_ERR_ARG;
    const std::string n(name);
    const double *src = nullptr;
    if (n == "rho") src = P->rho; else if (n == "p") src = P->p; else if (n == "p_rgh") src = P->p_rgh; else if (n == "T") src = P->T;
    else if (n == "h") src = P->hs; else if (n == "K") src = P->K; else if (n == "Ux") src = P->U[0]; else if (n == "Uy") src = P->U[1];
    else if (n == "Uz") src = P->U[2]; else if (n == "psi") src = P->psi; else if (n == "ph_rgh") src = P->ph_rgh;
    else if (n == "G") src = P->G; else if (n == "ShSu") src = P->radShSu; else if (n == "ShSp") src = P->radShSp; else if (n == "radE") src = P->radE;
    else if (n.size() > 1 && n[0] == 'I' && isdigit((unsigned char)n[1])) { const int i = atoi(n.c_str() + 1); if (i < (int)P->I.size()) src = P->I[i]; }
    else for (int i = 0; i < NSP; i++) if (n == SPN[i]) src = P->Y[i];
    if (!src) { ffm_set_error("unknown field %s", name); return FFM_ERR_ARG; }
    std::vector<double> v(P->N);
    PL_HIP(hipStreamSynchronize(P->ctx->stream));
    FFM_TRY(ffm_d2h(P->ctx, v.data(), src, sizeof(double) * P->N));
    for (int c = 0; c < P->nOwn; c++) out[P->newToOld[c]] = v[c];      // owned cells, local natural (blockMesh) order
    return FFM_OK;
}

// Raw state of the case for a driver that runs the SAME case through another path (bench.py: the reference's equation files over the
// Foam layer, libffm_refsnippets.so): cell fields [N] in the library's cell order, face fields [F] in the renumbered LDU face order
// (what ffm_faces_to_native takes), boundary arrays [B] in (patch, face) order -- inlet, floor, top, sides.  Returns the count.
extern "C" long ffm_plume_get_raw(ffm_plume *P, const char *name, double *out, long cap)
{
    if (!P || !name || !out) return FFM_ERR_ARG;
    if (P->N != P->nOwn) { ffm_set_error("ffm_plume_get_raw: single block only"); return FFM_ERR_ARG; }
    const std::string n(name);
    const long N = P->N, F = P->F, B = P->B;
    auto cellF = [&](const double *src) -> long { if (cap < N) return FFM_ERR_ARG; return ffm_d2h(P->ctx, out, src, sizeof(double) * N) == FFM_OK ? N : FFM_ERR_HIP; };
    auto bndF = [&](const double *src) -> long { if (cap < B) return FFM_ERR_ARG; return ffm_d2h(P->ctx, out, src, sizeof(double) * B) == FFM_OK ? B : FFM_ERR_HIP; };
    auto faceF = [&](const double *src) -> long {
        if (cap < F) return FFM_ERR_ARG;
        std::vector<double> v(std::max<long>(P->nNat, 1));
        if (ffm_d2h(P->ctx, v.data(), src, sizeof(double) * P->nNat) != FFM_OK) return FFM_ERR_HIP;
        const std::vector<int> &c2n = P->A->h_callerToNative;
        for (long f = 0; f < F; f++) out[f] = v[c2n[f]];
        return F;
    };
    if (n == "rho") return cellF(P->rho); if (n == "p") return cellF(P->p); if (n == "p_rgh") return cellF(P->p_rgh); if (n == "h") return cellF(P->hs);
    if (n == "K") return cellF(P->K); if (n == "dpdt") return cellF(P->dpdt); if (n == "gh") return cellF(P->gh); if (n == "T") return cellF(P->T);
    if (n == "Ux") return cellF(P->U[0]); if (n == "Uy") return cellF(P->U[1]); if (n == "Uz") return cellF(P->U[2]);
    for (int i = 0; i < NSP; i++) if (n == SPN[i]) return cellF(P->Y[i]);
    if (n == "phi") return faceF(P->phi); if (n == "ghf") return faceF(P->ghf);
    if (n == "phib") return bndF(P->phib); if (n == "ph_rgh_b") return bndF(P->ph_rgh_b); if (n == "kind") return bndF(P->kind_d);
    if (n == "fStaticS") return bndF(P->fStaticS); if (n == "fStaticH") return bndF(P->fStaticH); if (n == "refH") return bndF(P->refH);
    for (int c = 0; c < 3; c++) { if (n == std::string("fStaticU") + char('0' + c)) return bndF(P->fStaticU[c]); if (n == std::string("refU") + char('0' + c)) return bndF(P->refU[c]); }
    for (int i = 0; i < NSP; i++) if (n == std::string("refY") + char('0' + i)) return bndF(P->refY[i]);
    if (n == "ghfB") {      // gh on the boundary faces: the owner cell's gh moved by half a cell along the face normal
        if (cap < B) return FFM_ERR_ARG;
        std::vector<double> gh(N), sy(B), mg(B); std::vector<int> bc(B);
        if (ffm_d2h(P->ctx, gh.data(), P->gh, sizeof(double) * N) != FFM_OK || ffm_d2h(P->ctx, sy.data(), ffm_mesh_geom(P->mesh, 7), sizeof(double) * B) != FFM_OK ||
            ffm_d2h(P->ctx, mg.data(), ffm_mesh_geom(P->mesh, 4), sizeof(double) * B) != FFM_OK || ffm_d2h(P->ctx, bc.data(), ffm_mesh_bcells(P->mesh), sizeof(int) * B) != FFM_OK) return FFM_ERR_HIP;
        for (long k = 0; k < B; k++) out[k] = gh[bc[k]] + (-9.81) * (0.5 * P->h * (sy[k] / mg[k]));
        return B;
    }
    ffm_set_error("ffm_plume_get_raw: unknown array %s", name);
    return FFM_ERR_ARG;
}

extern "C" int ffm_plume_nsolves(const ffm_plume *P) { return P ? (int)P->log.size() : FFM_ERR_ARG; }
extern "C" int ffm_plume_get_solve(const ffm_plume *P, int i, char *name16, ffm_perf *perf)
{
    if (!P || i < 0 || i >= (int)P->log.size()) return FFM_ERR_ARG;
    if (name16) memcpy(name16, P->log[i].name, 16);
    if (perf) *perf = P->log[i].perf;
    return FFM_OK;
}
extern "C" ffm_ldu *ffm_plume_ldu(ffm_plume *P) { return P ? P->A : nullptr; }
extern "C" ffm_mesh *ffm_plume_mesh(ffm_plume *P) { return P ? P->mesh : nullptr; }
