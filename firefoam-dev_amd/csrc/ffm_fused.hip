// ffm_fused.hip -- fused assembly passes of the compiled time step (ffm_plume.hip).
//
// The per-operator kernels of ffm_fv.hip mirror OpenFOAM's one-pass-per-operator evaluation (what the Foam layer in
// include/ffmFoam.H calls, operator by operator).  A transport equation assembled that way costs six passes over the mesh
// (limiter weights, coefficients, boundary coefficients, ddt source, explicit source, addBoundaryDiag/Source), and the
// face geometry and addressing are streamed again by each of them.  The kernels here produce the SAME numbers -- every
// expression is the one of the per-operator kernel it replaces, evaluated in the same order, FMA contraction off -- in one
// pass per equation, several equations that share phi / rho / the diffusivity (the species of solver/YEEqn.H:37-67) per launch:
//   k_grad_multi        fvc::grad of NF fields in one pass (Gauss linear; solver/YEEqn.H:8 limiters, solver/UEqn.H:5 LUST)
//   k_scalar_eqns       fvm::ddt(rho,Yi) + mvConvection->fvmDiv(phi,Yi) - fvm::laplacian(dEff,Yi) == Su  for NF fields:
//                       limitedLinear / limitedLinear01 weights on the fly (NVDTVD::r), matrix coefficients, boundary
//                       coefficients of the mixed patch condition, source with the explicit terms, addBoundaryDiag/Source
//   k_lust_source       the explicit part of `div(phi,U) Gauss LUST grad(U)` (cases/steckler/system/fvSchemes:32) for the
//                       three components + the ddt source:  source_c = rDeltaT*rho0*U0_c*V - V*surfaceIntegrate(phi*corr_c)
// tests/test_fused_gpu.py checks each of them bitwise against the chain of per-operator entry points.
#include "ffm_mesh.hpp"

constexpr int FUSE_MAX = 4;
#define CHECK_M(m) if (!(m)) { ffm_set_error("null mesh"); return FFM_ERR_ARG; }

struct GradMulti {
    const double *vf[FUSE_MAX], *vb[FUSE_MAX];
    double *gx[FUSE_MAX], *gy[FUSE_MAX], *gz[FUSE_MAX];
};

// fvc::grad, Gauss linear, of NF fields: one pass over the addressing and the face geometry (k_grad of ffm_fv.hip per field)
template <int W, int NF>
__global__ __launch_bounds__(256) void k_grad_multi(MeshView q, GradMulti a)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci;
        RowEnt<W> L, U; load_lower<W>(q.v, c, L); load_upper<W>(q.v, c, U);
        double wl[W], wu[W], sx[2 * W], sy[2 * W], sz[2 * W];
#pragma unroll
        for (int s = 0; s < W; s++) {
            wl[s] = q.w[L.f[s]]; wu[s] = q.w[U.f[s]];
            sx[s] = q.Sfx[L.f[s]]; sy[s] = q.Sfy[L.f[s]]; sz[s] = q.Sfz[L.f[s]];
            sx[W + s] = q.Sfx[U.f[s]]; sy[W + s] = q.Sfy[U.f[s]]; sz[W + s] = q.Sfz[U.f[s]];
        }
        const double V = q.V[c];
        const int j = q.cellB[c];
#pragma unroll 1            // (fields one after the other: 9 % faster than interleaving their loads, measured r02)
        for (int i = 0; i < NF; i++) {
            const double *__restrict__ vf = a.vf[i];
            const double P = vf[c];
            double vl[W], vu[W];
#pragma unroll
            for (int s = 0; s < W; s++) { vl[s] = vf[L.nb[s]]; vu[s] = vf[U.nb[s]]; }
            double ax = 0, ay = 0, az = 0;
#pragma unroll
            for (int s = 0; s < W; s++) if (L.on[s]) {
                const double ff = wl[s] * vl[s] + (1.0 - wl[s]) * P;      // owner of this face is the neighbour cell
                ax -= sx[s] * ff; ay -= sy[s] * ff; az -= sz[s] * ff;
            }
#pragma unroll
            for (int s = 0; s < W; s++) if (U.on[s]) {
                const double ff = wu[s] * P + (1.0 - wu[s]) * vu[s];
                ax += sx[W + s] * ff; ay += sy[W + s] * ff; az += sz[W + s] * ff;
            }
            if (j >= 0) for (int t = q.bcStart[j]; t < q.bcStart[j + 1]; t++) {
                const int k = q.bcItem[t]; const double b = a.vb[i][k];
                ax += q.bSfx[k] * b; ay += q.bSfy[k] * b; az += q.bSfz[k] * b;
            }
            a.gx[i][c] = ax / V; a.gy[i][c] = ay / V; a.gz[i][c] = az / V;
        }
    }
}

// limitedLinear(k) / limitedLinear01(k) weight of one face (k_limited_weights of ffm_fv.hip): o = owner, n = neighbour
__device__ __forceinline__ double limited_weight(int scheme, double twoByk, double lo, double hi, double flux, double wlin,
                                                 double P, double Nn, double dx, double dy, double dz, double gxu, double gyu, double gzu)
{
    const double p0 = flux >= 0 ? 1.0 : 0.0;
    const double gradf = Nn - P;
    const double gradcf = dx * gxu + dy * gyu + dz * gzu;
    double r;
    if (fabs(gradcf) >= 1000.0 * fabs(gradf)) {
        const double sa = gradcf >= 0 ? 1.0 : -1.0, sb = gradf >= 0 ? 1.0 : -1.0;
        r = 2.0 * 1000.0 * sa * sb - 1.0;
    } else r = 2.0 * (gradcf / gradf) - 1.0;
    double lim = fmax(fmin(twoByk * r, 1.0), 0.0);
    if (scheme == 3) {
        if ((flux > 0 && (P < lo || Nn > hi)) || (flux < 0 && (Nn < lo || P > hi))) lim = 0.0;
    }
    return lim * wlin + (1.0 - lim) * p0;
}

struct ScalarEqns {
    // per field
    const double *vf[FUSE_MAX], *gx[FUSE_MAX], *gy[FUSE_MAX], *gz[FUSE_MAX], *vf0[FUSE_MAX];
    const double *f[FUSE_MAX], *ref[FUSE_MAX], *refGrad[FUSE_MAX];      // mixed patch condition [B]
    const double *su[FUSE_MAX];                                          // explicit volume source (nullable)
    const double *su2[FUSE_MAX], *sp[FUSE_MAX];                          // a second explicit source and an implicit one (fvm::Sp): nullable
    const double *expl[FUSE_MAX][3];                                     // explicit LHS volume terms (nullable)
    double *diag[FUSE_MAX], *upper[FUSE_MAX], *lower[FUSE_MAX], *src[FUSE_MAX];
    // shared
    const double *rho, *rho0, *phi, *phib, *gamma, *gammab;
    const double *Cx, *Cy, *Cz, *bMagSf, *bDelta;
    double rdt, twoByk, lo, hi;
    int scheme;                                                          // 2 limitedLinear, 3 limitedLinear01
    const double *wGiven;                                                // GIVENW: the face weights (a multivariate scheme's common ones)
    int nOffDiag;                                                        // GIVENW: fields 0 .. nOffDiag-1 write their off-diagonals (with common weights and one
                                                                         // diffusivity every field's are the same: the caller shares field 0's, or an earlier pass')
};

// Occupancy: the fused four-field kernel is bound by the latency of its neighbour gathers; capped at 128 VGPRs (4 waves per SIMD,
// 84 bytes of scratch per lane) it takes 17.5 ms at 400^3 instead of 21.9 ms with the 150 VGPRs the compiler picks by itself; 5 / 6
// waves spill too much (25 / 27 ms), and the one-field kernel is better left alone (7.97 vs 8.35 ms).  Measured r02u.
// GIVENW: the face weights come from a.wGiven (a multivariateSelection scheme's common weights) instead of one limiter per field: no
// gradient / neighbour-value gathers at all
template <int W, int NF, bool GIVENW = false>
__global__ __launch_bounds__(256, ((NF > 1 && !GIVENW) ? 4 : 1)) void k_scalar_eqns(MeshView q, ScalarEqns a)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci;
        RowEnt<W> L, U; load_lower<W>(q.v, c, L); load_upper<W>(q.v, c, U);
        // shared face data: flux, linear weight, laplacian coefficient gamma*magSf*delta, distance vector
        double fl[W], fu[W], wl[W], wu[W], gl[W], gu[W], dlx[W], dly[W], dlz[W], dux[W], duy[W], duz[W];
        const double cx = GIVENW ? 0.0 : a.Cx[c], cy = GIVENW ? 0.0 : a.Cy[c], cz = GIVENW ? 0.0 : a.Cz[c];
#pragma unroll
        for (int s = 0; s < W; s++) {
            const int el = L.f[s], eu = U.f[s];
            fl[s] = a.phi[el]; fu[s] = a.phi[eu];
            wl[s] = GIVENW ? a.wGiven[el] : q.w[el]; wu[s] = GIVENW ? a.wGiven[eu] : q.w[eu];
            gl[s] = a.gamma[el] * q.magSf[el] * q.delta[el]; gu[s] = a.gamma[eu] * q.magSf[eu] * q.delta[eu];
            if (!GIVENW) {      // d = C[neighbour] - C[owner]
                dlx[s] = cx - a.Cx[L.nb[s]]; dly[s] = cy - a.Cy[L.nb[s]]; dlz[s] = cz - a.Cz[L.nb[s]];
                dux[s] = a.Cx[U.nb[s]] - cx; duy[s] = a.Cy[U.nb[s]] - cy; duz[s] = a.Cz[U.nb[s]] - cz;
            } else { dlx[s] = dly[s] = dlz[s] = dux[s] = duy[s] = duz[s] = 0.0; }
        }
        const double V = q.V[c], rhoc = a.rho[c], rho0c = a.rho0[c];
        const int j = q.cellB[c];
#ifndef FFM_SE2_UNROLL
#define FFM_SE2_UNROLL 1
#endif
#pragma unroll FFM_SE2_UNROLL
        for (int i = 0; i < NF; i++) {
            const double *__restrict__ vf = a.vf[i], *__restrict__ gx = a.gx[i], *__restrict__ gy = a.gy[i], *__restrict__ gz = a.gz[i];
            const double P = GIVENW ? 0.0 : vf[c], gxc = GIVENW ? 0.0 : gx[c], gyc = GIVENW ? 0.0 : gy[c], gzc = GIVENW ? 0.0 : gz[c];
            double dDiv = 0.0, dLap = 0.0;
            // faces where c is the neighbour (owner = L.nb[s]): diag -= upper[f]
#pragma unroll
            for (int s = 0; s < W; s++) if (L.on[s]) {
                const int o = L.nb[s];
                const double flux = fl[s];
                const bool upO = flux > 0;                       // upwind cell: the owner when the flux is positive
                const double w = GIVENW ? wl[s] : limited_weight(a.scheme, a.twoByk, a.lo, a.hi, flux, wl[s], vf[o], P, dlx[s], dly[s], dlz[s],
                                                                  upO ? gx[o] : gxc, upO ? gy[o] : gyc, upO ? gz[o] : gzc);
                (void)o; (void)upO;
                const double lo = -w * flux;
                dDiv -= (lo + flux);
                dLap -= gl[s];
            }
            // faces owned by c: write the coefficients, diag -= lower[f]
#pragma unroll
            for (int s = 0; s < W; s++) if (U.on[s]) {
                const int n = U.nb[s];
                const double flux = fu[s];
                const bool upO = flux > 0;
                const double w = GIVENW ? wu[s] : limited_weight(a.scheme, a.twoByk, a.lo, a.hi, flux, wu[s], P, vf[n], dux[s], duy[s], duz[s],
                                                                  upO ? gxc : gx[n], upO ? gyc : gy[n], upO ? gzc : gz[n]);
                (void)n; (void)upO;
                double lo = -w * flux, up = lo + flux;
                dDiv -= lo;
                dLap -= gu[s];
                lo = lo - gu[s]; up = up - gu[s];
                if (!GIVENW || i < a.nOffDiag) { a.upper[i][U.f[s]] = up; a.lower[i][U.f[s]] = lo; }
            }
            double d = a.rdt * rhoc * V;
            d = d + dDiv;
            d = d - dLap;
            // source: ddt, explicit LHS terms (one `source -= V*term` each), explicit source
            double sc = a.rdt * rho0c * a.vf0[i][c] * V;
            if (a.expl[i][0]) sc = ((sc - V * a.expl[i][0][c]) - V * a.expl[i][1][c]) - V * a.expl[i][2][c];
            if (a.su[i]) sc = sc + V * a.su[i][c];
            if (a.sp[i]) d = d + V * a.sp[i][c];
            if (a.su2[i]) sc = sc + V * a.su2[i][c];
            // boundary coefficients of the mixed condition, added in (patch, face) order
            if (j >= 0) for (int t = q.bcStart[j]; t < q.bcStart[j + 1]; t++) {
                const int k = q.bcItem[t];
                const double fk = a.f[i][k], rk = a.ref[i][k], gk = a.refGrad[i][k], dk = a.bDelta[k], pb = a.phib[k];
                double ic = pb * (1.0 - fk), bc = -pb * (fk * rk + (1.0 - fk) * gk / dk);
                const double pG = a.gammab[k] * a.bMagSf[k];
                const double li = pG * (-fk * dk), lb = -pG * (fk * dk * rk + (1.0 - fk) * gk);
                ic = ic - li; bc = bc - lb;
                d += ic; sc += bc;
            }
            a.diag[i][c] = d; a.src[i][c] = sc;
        }
    }
}

// multivariateSelectionScheme (solver/YEEqn.H:1-10; cases/steckler/system/fvSchemes:36-47): the weights of ONE limiter for all fields
// of the table -- the face-wise minimum of limitedLinear(01) limiters of NF fields (k_limited_weights<1/2> of ffm_fv.hip per field and
// k_weights_from_limiter, in one pass over the owned faces)
constexpr int MV_MAX = 8;
struct MvWeights {
    const double *vf[MV_MAX], *gx[MV_MAX], *gy[MV_MAX], *gz[MV_MAX];
    int scheme[MV_MAX], nf;
    const double *phi, *Cx, *Cy, *Cz;
    double twoByk, lo, hi;
    double *out;
};
// One limiter value of one field: limited_weight() with linear weight 1 and upwind weight 0
__device__ __forceinline__ double mv_limiter(int scheme, double twoByk, double lo, double hi, double flux, double P, double Nn, double gradcf)
{
    const double gradf = Nn - P;
    double r;
    if (fabs(gradcf) >= 1000.0 * fabs(gradf)) {
        const double sa = gradcf >= 0 ? 1.0 : -1.0, sb = gradf >= 0 ? 1.0 : -1.0;
        r = 2.0 * 1000.0 * sa * sb - 1.0;
    } else r = 2.0 * (gradcf / gradf) - 1.0;
    double l = fmax(fmin(twoByk * r, 1.0), 0.0);
    if (scheme == 3) {
        if ((flux > 0 && (P < lo || Nn > hi)) || (flux < 0 && (Nn < lo || P > hi))) l = 0.0;
    }
    return l;
}
// Every face is written by its UPWIND cell (the owner when the flux is positive, else the neighbour): the limiter needs the upwind
// cell's gradient of every field, which that cell holds itself -- only the other cell's value is gathered (8 B per field and face
// instead of 32).  P / N keep their owner / neighbour meaning, so the arithmetic is that of the owner-face loop, bit for bit.
template <int W>
__global__ __launch_bounds__(256) void k_mv_weights(MeshView q, MvWeights a)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci;
        RowEnt<W> L, U; load_lower<W>(q.v, c, L); load_upper<W>(q.v, c, U);
        const double cx = a.Cx[c], cy = a.Cy[c], cz = a.Cz[c];
        double fl[2 * W], lim[2 * W], dx[2 * W], dy[2 * W], dz[2 * W];
        bool mine[2 * W], ghostUp[W];       // ghostUp: the upwind cell is a ghost cell (a cut face of a decomposed mesh, owned by c): c writes it too
#pragma unroll
        for (int s = 0; s < W; s++) {
            // lower face: owner = L.nb[s], neighbour = c; upwind is c when the flux is not positive.  d = C[neighbour] - C[owner]
            fl[s] = L.on[s] ? a.phi[L.f[s]] : 0.0; mine[s] = L.on[s] && !(fl[s] > 0);
            dx[s] = cx - a.Cx[L.nb[s]]; dy[s] = cy - a.Cy[L.nb[s]]; dz[s] = cz - a.Cz[L.nb[s]];
            // upper face: owner = c, neighbour = U.nb[s]; upwind is c when the flux is positive
            fl[W + s] = U.on[s] ? a.phi[U.f[s]] : 0.0; mine[W + s] = U.on[s] && fl[W + s] > 0;
            ghostUp[s] = U.on[s] && !(fl[W + s] > 0) && U.nb[s] >= q.v.N;
            dx[W + s] = a.Cx[U.nb[s]] - cx; dy[W + s] = a.Cy[U.nb[s]] - cy; dz[W + s] = a.Cz[U.nb[s]] - cz;
            lim[s] = lim[W + s] = 1.0;
        }
#pragma unroll 1
        for (int i = 0; i < a.nf; i++) {
            const double *__restrict__ vf = a.vf[i];
            const double vc = vf[c], gxc = a.gx[i][c], gyc = a.gy[i][c], gzc = a.gz[i][c];
            const int sch = a.scheme[i];
#pragma unroll
            for (int s = 0; s < W; s++) {
                if (mine[s]) {          // P = owner's value (the other cell), N = c's
                    const double l = mv_limiter(sch, a.twoByk, a.lo, a.hi, fl[s], vf[L.nb[s]], vc, dx[s] * gxc + dy[s] * gyc + dz[s] * gzc);
                    lim[s] = i == 0 ? l : fmin(lim[s], l);
                }
                if (mine[W + s]) {      // P = c's value, N = the other cell's
                    const double l = mv_limiter(sch, a.twoByk, a.lo, a.hi, fl[W + s], vc, vf[U.nb[s]], dx[W + s] * gxc + dy[W + s] * gyc + dz[W + s] * gzc);
                    lim[W + s] = i == 0 ? l : fmin(lim[W + s], l);
                } else if (ghostUp[s]) {
                    const int n = U.nb[s];
                    const double l = mv_limiter(sch, a.twoByk, a.lo, a.hi, fl[W + s], vc, vf[n], dx[W + s] * a.gx[i][n] + dy[W + s] * a.gy[i][n] + dz[W + s] * a.gz[i][n]);
                    lim[W + s] = i == 0 ? l : fmin(lim[W + s], l);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < W; s++) {
            if (mine[s]) { const int e = L.f[s]; const double p0 = fl[s] >= 0 ? 1.0 : 0.0; a.out[e] = lim[s] * q.w[e] + (1.0 - lim[s]) * p0; }
            if (mine[W + s] || ghostUp[s]) { const int e = U.f[s]; const double p0 = fl[W + s] >= 0 ? 1.0 : 0.0; a.out[e] = lim[W + s] * q.w[e] + (1.0 - lim[W + s]) * p0; }
        }
    }
}

// ---- the same weights with the cell fields staged through LDS on the tile numbering -------------------------------
// k_grad_multi (the Gauss gradients of the nf fields) + k_mv_weights in ONE pass, bit for bit: a workgroup walks a run of consecutive
// entries (dependency levels of at most 256 cells) of one tile; the nf cell values of the entries e-1, e, e+1 sit in an LDS window
// (three buffers, entry e+1 loaded while entry e is computed), so the six neighbours of a cell -- which on the tile numbering lie in
// the entries next to its own -- are served from LDS: every cell value is read from memory once per run instead of once by the
// cell and once by each neighbour.  A cell forms its own gradients in registers (the limiter of a face needs the
// UPWIND cell's gradient, and every face is written by its upwind cell), so the eighteen gradient arrays are neither written nor read.
// Neighbours outside the window (other tiles: ~12 % of the faces) are read from memory.  Single block only (no ghost cells).
// Measured at 400^3 (profiles/r03*): 11.8-12.7 ms against 17.3 ms for the three passes it replaces (k_grad_multi<3,2> + <3,4> + k_mv_weights);
// the pass is then bound by its ~5 000 instructions per cell (54 fp64 divisions of the gradients and of NVDTVD::r), not by traffic: holding the
// centres in LDS as well (217 VGPRs, 2 waves / SIMD) or forcing 3 waves / SIMD (168 VGPRs, 140 B of scratch) changes it by < 4 %.
constexpr int MVT_ENT = 256, MVT_FLD = 6, MVT_RUN = 32;
struct MvTile {
    const double *vf[MVT_FLD], *vb[MVT_FLD];
    int scheme[MVT_FLD], nf;
    const double *phi, *Cx, *Cy, *Cz;
    double twoByk, lo, hi;
    double *out;
    const int4 *seg, *rec;
};
template <int W>
__global__ __launch_bounds__(256) void k_mv_tile(MeshView q, MvTile a)
{
    __shared__ double ring[MVT_FLD][3 * MVT_ENT];
    __shared__ int4 shRec[3];
    const int4 sg = a.seg[blockIdx.x];
    const int gBeg = sg.x, ea = sg.y, eb = sg.z, gEnd = sg.w;
    const int t = threadIdx.x;
    const int nf = a.nf;
    auto stage = [&](int ee) {          // cell values + centres of entry ee -> buffer ee % 3
        if (ee < gBeg || ee >= gEnd) { if (t == 0) shRec[(ee + 3) % 3] = make_int4(0, 0, 0, 0); return; }
        const int4 R = a.rec[ee];
        const int c0 = R.x, cnt = R.y & 0xFFFF;
        if (t == 0) shRec[ee % 3] = make_int4(c0, cnt, 0, 0);
        if (t < cnt) {
            const int c = c0 + t, o = (ee % 3) * MVT_ENT + t;
            for (int i = 0; i < nf; i++) ring[i][o] = a.vf[i][c];
        }
    };
    stage(ea - 1); stage(ea);
    for (int ee = ea; ee < eb; ee++) {
        stage(ee + 1);
        __syncthreads();
        const int4 Rm = shRec[(ee + 2) % 3], R0 = shRec[ee % 3], Rp = shRec[(ee + 1) % 3];
        if (t < R0.y) {
            const int c = R0.x + t, own = (ee % 3) * MVT_ENT + t;
            RowEnt<W> L, U; load_lower<W>(q.v, c, L); load_upper<W>(q.v, c, U);
            // LDS slot of a neighbour cell, or -1: outside the window
            auto slotOf = [&](int nb) -> int {
                if (nb >= Rm.x && nb < Rm.x + Rm.y) return ((ee + 2) % 3) * MVT_ENT + (nb - Rm.x);
                if (nb >= Rp.x && nb < Rp.x + Rp.y) return ((ee + 1) % 3) * MVT_ENT + (nb - Rp.x);
                return -1;
            };
            const double cx = a.Cx[c], cy = a.Cy[c], cz = a.Cz[c];
            double fl[2 * W], lim[2 * W], dx[2 * W], dy[2 * W], dz[2 * W], wl[W], wu[W], sx[2 * W], sy[2 * W], sz[2 * W];
            int sl[2 * W], gn[2 * W];       // LDS slot of the neighbour (-1: outside the window) and the cell to load from memory in that case
            bool mine[2 * W];
            // Neighbours outside the window (other tiles) come from memory.  Their loads are issued for EVERY lane -- the lanes whose
            // neighbour is in LDS load their own cell, a cache hit -- so that no load sits in a divergent branch: six independent loads
            // per field, the next field's issued before this field's arithmetic (a dependent chain of 36 exposed latencies otherwise).
#pragma unroll
            for (int s = 0; s < W; s++) {
                sl[s] = L.on[s] ? slotOf(L.nb[s]) : own; sl[W + s] = U.on[s] ? slotOf(U.nb[s]) : own;
                gn[s] = sl[s] >= 0 ? c : L.nb[s]; gn[W + s] = sl[W + s] >= 0 ? c : U.nb[s];
            }
            double gcur[2 * W];
#pragma unroll
            for (int s = 0; s < 2 * W; s++) gcur[s] = a.vf[0][gn[s]];
#pragma unroll
            for (int s = 0; s < W; s++) {
                // lower face: owner = L.nb[s], neighbour = c; upwind is c when the flux is not positive.  d = C[neighbour] - C[owner]
                fl[s] = L.on[s] ? a.phi[L.f[s]] : 0.0; mine[s] = L.on[s] && !(fl[s] > 0);
                fl[W + s] = U.on[s] ? a.phi[U.f[s]] : 0.0; mine[W + s] = U.on[s] && fl[W + s] > 0;
                lim[s] = lim[W + s] = 1.0;
                // face geometry of the gradient (k_grad_multi)
                wl[s] = q.w[L.f[s]]; wu[s] = q.w[U.f[s]];
                sx[s] = q.Sfx[L.f[s]]; sy[s] = q.Sfy[L.f[s]]; sz[s] = q.Sfz[L.f[s]];
                sx[W + s] = q.Sfx[U.f[s]]; sy[W + s] = q.Sfy[U.f[s]]; sz[W + s] = q.Sfz[U.f[s]];
            }
#pragma unroll
            for (int s = 0; s < W; s++) {       // d = C[neighbour] - C[owner]; the cell centres of the neighbours are gathered (cache hits inside a tile)
                dx[s] = cx - a.Cx[L.nb[s]]; dy[s] = cy - a.Cy[L.nb[s]]; dz[s] = cz - a.Cz[L.nb[s]];
                dx[W + s] = a.Cx[U.nb[s]] - cx; dy[W + s] = a.Cy[U.nb[s]] - cy; dz[W + s] = a.Cz[U.nb[s]] - cz;
            }
            const double V = q.V[c];
            const int j = q.cellB[c];
#pragma unroll 1
            for (int i = 0; i < nf; i++) {
                const double P = ring[i][own];
                double vl[W], vu[W], gnext[2 * W];
                const double *__restrict__ vfn = a.vf[i + 1 < nf ? i + 1 : i];
#pragma unroll
                for (int s = 0; s < 2 * W; s++) gnext[s] = vfn[gn[s]];            // the next field's values from memory: in flight during this field's arithmetic
#pragma unroll
                for (int s = 0; s < W; s++) {
                    const int k = sl[s] >= 0 ? sl[s] : own, ku = sl[W + s] >= 0 ? sl[W + s] : own;
                    const double rl = ring[i][k], ru = ring[i][ku];
                    vl[s] = sl[s] >= 0 ? rl : gcur[s]; vu[s] = sl[W + s] >= 0 ? ru : gcur[W + s];
                }
                // ---- fvc::grad of field i at c (k_grad_multi, same order: lower faces, upper faces, boundary faces)
                double ax = 0, ay = 0, az = 0;
#pragma unroll
                for (int s = 0; s < W; s++) if (L.on[s]) {
                    const double ff = wl[s] * vl[s] + (1.0 - wl[s]) * P;
                    ax -= sx[s] * ff; ay -= sy[s] * ff; az -= sz[s] * ff;
                }
#pragma unroll
                for (int s = 0; s < W; s++) if (U.on[s]) {
                    const double ff = wu[s] * P + (1.0 - wu[s]) * vu[s];
                    ax += sx[W + s] * ff; ay += sy[W + s] * ff; az += sz[W + s] * ff;
                }
                if (j >= 0) for (int tt = q.bcStart[j]; tt < q.bcStart[j + 1]; tt++) {
                    const int k = q.bcItem[tt]; const double b = a.vb[i][k];
                    ax += q.bSfx[k] * b; ay += q.bSfy[k] * b; az += q.bSfz[k] * b;
                }
                const double gxc = ax / V, gyc = ay / V, gzc = az / V;
                // ---- the limiter of field i on the faces whose upwind cell is c (k_mv_weights)
                const int sch = a.scheme[i];
#pragma unroll
                for (int s = 0; s < W; s++) {
                    if (mine[s]) {          // P = owner's value (the other cell), N = c's
                        const double l = mv_limiter(sch, a.twoByk, a.lo, a.hi, fl[s], vl[s], P, dx[s] * gxc + dy[s] * gyc + dz[s] * gzc);
                        lim[s] = i == 0 ? l : fmin(lim[s], l);
                    }
                    if (mine[W + s]) {      // P = c's value, N = the other cell's
                        const double l = mv_limiter(sch, a.twoByk, a.lo, a.hi, fl[W + s], P, vu[s], dx[W + s] * gxc + dy[W + s] * gyc + dz[W + s] * gzc);
                        lim[W + s] = i == 0 ? l : fmin(lim[W + s], l);
                    }
                }
#pragma unroll
                for (int s = 0; s < 2 * W; s++) gcur[s] = gnext[s];
            }
#pragma unroll
            for (int s = 0; s < W; s++) {
                if (mine[s]) { const int e = L.f[s]; const double p0 = fl[s] >= 0 ? 1.0 : 0.0; a.out[e] = lim[s] * q.w[e] + (1.0 - lim[s]) * p0; }
                if (mine[W + s]) { const int e = U.f[s]; const double p0 = fl[W + s] >= 0 ? 1.0 : 0.0; a.out[e] = lim[W + s] * q.w[e] + (1.0 - lim[W + s]) * p0; }
            }
        }
        __syncthreads();            // entry ee is done: its e-1 buffer may be overwritten by the next stage()
    }
}

// ffm_fvc_grad_multi over the nf fields + ffm_fv_multivariate_weights in one LDS-staged pass (bit for bit); FFM_ERR_UNSUPPORTED where the
// mesh has no tile plan / is one block of a decomposed mesh / nf > 6 -- the caller then runs the two-pass form
extern "C" int ffm_fv_multivariate_weights_tiled(ffm_mesh *m, int nf, const int *schemes, double k, double lo, double hi, const double *phi_f,
                                                 const double *const *vf, const double *const *vb, double *out_w)
{
    CHECK_M(m);
    if (nf < 1 || !schemes || !phi_f || !vf || !vb || !out_w) return FFM_ERR_ARG;
    FfmFvSegs sg;
    if (nf > MVT_FLD || m->A->maxW > 3 || getenv("FFM_NO_TILE_FV") || !ffm_tile_fv_segments(m->A, MVT_RUN, &sg)) return FFM_ERR_UNSUPPORTED;
    MvTile a;
    for (int i = 0; i < MVT_FLD; i++) {
        const int qq = i < nf ? i : 0;
        if (i < nf && ((schemes[qq] != 2 && schemes[qq] != 3) || !vf[qq] || (m->B && !vb[qq]))) return FFM_ERR_ARG;
        a.vf[i] = vf[qq]; a.vb[i] = vb[qq]; a.scheme[i] = schemes[qq];
    }
    a.nf = nf; a.phi = phi_f; a.Cx = m->C[0]; a.Cy = m->C[1]; a.Cz = m->C[2]; a.twoByk = 2.0 / std::max(k, 1e-15); a.lo = lo; a.hi = hi; a.out = out_w;
    a.seg = sg.seg; a.rec = sg.rec;
    hipLaunchKernelGGL(k_mv_tile<3>, dim3(sg.nSeg), dim3(256), 0, m->ctx->stream, mview(m), a);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

struct LustSource {
    const double *gx[3], *gy[3], *gz[3], *U0[3];
    double *src[3];
    const double *phi, *rho0, *Cx, *Cy, *Cz, *Cfx, *Cfy, *Cfz;
    double rdt;
};

template <int W>
__global__ __launch_bounds__(256) void k_lust_source(MeshView q, LustSource a)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci;
        RowEnt<W> L, U; load_lower<W>(q.v, c, L); load_upper<W>(q.v, c, U);
        const double cx = a.Cx[c], cy = a.Cy[c], cz = a.Cz[c];
        double fl[W], fu[W], dlx[W], dly[W], dlz[W], dux[W], duy[W], duz[W];
        int upl[W], upu[W];
#pragma unroll
        for (int s = 0; s < W; s++) {
            const int el = L.f[s], eu = U.f[s];
            fl[s] = a.phi[el]; fu[s] = a.phi[eu];
            upl[s] = fl[s] > 0 ? L.nb[s] : c;           // lower face: owner = L.nb[s], neighbour = c
            upu[s] = fu[s] > 0 ? c : U.nb[s];
            dlx[s] = a.Cfx[el] - a.Cx[upl[s]]; dly[s] = a.Cfy[el] - a.Cy[upl[s]]; dlz[s] = a.Cfz[el] - a.Cz[upl[s]];
            dux[s] = a.Cfx[eu] - a.Cx[upu[s]]; duy[s] = a.Cfy[eu] - a.Cy[upu[s]]; duz[s] = a.Cfz[eu] - a.Cz[upu[s]];
        }
        (void)cx; (void)cy; (void)cz;
        const double V = q.V[c], r0 = a.rho0[c];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const double *__restrict__ gx = a.gx[i], *__restrict__ gy = a.gy[i], *__restrict__ gz = a.gz[i];
            double acc = 0.0;
#pragma unroll
            for (int s = 0; s < W; s++) if (L.on[s]) {
                const int u = upl[s];
                const double corr = 0.25 * ((dlx[s] * gx[u] + dly[s] * gy[u]) + dlz[s] * gz[u]);
                acc -= fl[s] * corr;
            }
#pragma unroll
            for (int s = 0; s < W; s++) if (U.on[s]) {
                const int u = upu[s];
                const double corr = 0.25 * ((dux[s] * gx[u] + duy[s] * gy[u]) + duz[s] * gz[u]);
                acc += fu[s] * corr;
            }
            const double divc = acc / V;
            a.src[i][c] = a.rdt * r0 * a.U0[i][c] * V - V * divc;
        }
    }
}

// ------------------------------------------------------------------ entry points (internal + C ABI) ---

extern "C" int ffm_fvc_grad_multi(ffm_mesh *m, int nf, const double *const *vf, const double *const *vb, double *const *gx,
                                  double *const *gy, double *const *gz)
{
    CHECK_M(m);
    if (nf < 1 || nf > FUSE_MAX || !vf || !vb || !gx || !gy || !gz) return FFM_ERR_ARG;
    GradMulti a;
    for (int i = 0; i < FUSE_MAX; i++) { const int k = i < nf ? i : 0; a.vf[i] = vf[k]; a.vb[i] = vb[k]; a.gx[i] = gx[k]; a.gy[i] = gy[k]; a.gz[i] = gz[k]; }
    for (int i = 0; i < nf; i++) if (!vf[i] || (m->B && !vb[i]) || !gx[i] || !gy[i] || !gz[i]) return FFM_ERR_ARG;
#define GM(NF) FFM_DISPATCH_W(m->A->maxW, LAUNCH_CELLS((k_grad_multi<W, NF>), mview(m), a))
    switch (nf) { case 1: GM(1); break; case 2: GM(2); break; case 3: GM(3); break; default: GM(4); break; }
#undef GM
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

extern "C" int ffm_fvm_scalar_transport_multi(ffm_mesh *m, int nf, int scheme, double k, double lo, double hi, double rDeltaT,
                                              const double *rho, const double *rho0, const double *phi_f, const double *phi_b,
                                              const double *gamma_f, const double *gamma_b,
                                              const double *const *vf, const double *const *gx, const double *const *gy,
                                              const double *const *gz, const double *const *vf0, const double *const *f,
                                              const double *const *ref, const double *const *refGrad, const double *const *su,
                                              const double *const *su2, const double *const *sp,
                                              const double *const *expl3, double *const *diag, double *const *upper,
                                              double *const *lower, double *const *source)
{
    CHECK_M(m);
    if (nf < 1 || nf > FUSE_MAX || (scheme != 2 && scheme != 3) || !rho || !rho0 || !phi_f || !gamma_f || (m->B && (!phi_b || !gamma_b)))
        return FFM_ERR_ARG;
    if (!vf || !gx || !gy || !gz || !vf0 || !f || !ref || !refGrad || !diag || !upper || !lower || !source) return FFM_ERR_ARG;
    ScalarEqns a;
    for (int i = 0; i < FUSE_MAX; i++) {
        const int q = i < nf ? i : 0;
        if (i < nf && (!vf[q] || !gx[q] || !gy[q] || !gz[q] || !vf0[q] || !diag[q] || !upper[q] || !lower[q] || !source[q] ||
                       (m->B && (!f[q] || !ref[q] || !refGrad[q])))) return FFM_ERR_ARG;
        a.vf[i] = vf[q]; a.gx[i] = gx[q]; a.gy[i] = gy[q]; a.gz[i] = gz[q]; a.vf0[i] = vf0[q];
        a.f[i] = f[q]; a.ref[i] = ref[q]; a.refGrad[i] = refGrad[q];
        a.su[i] = su ? su[q] : nullptr;
        a.su2[i] = su2 ? su2[q] : nullptr; a.sp[i] = sp ? sp[q] : nullptr;
        for (int e = 0; e < 3; e++) a.expl[i][e] = expl3 ? expl3[3 * q + e] : nullptr;
        if (a.expl[i][0] && (!a.expl[i][1] || !a.expl[i][2])) return FFM_ERR_ARG;
        a.diag[i] = diag[q]; a.upper[i] = upper[q]; a.lower[i] = lower[q]; a.src[i] = source[q];
    }
    a.rho = rho; a.rho0 = rho0; a.phi = phi_f; a.phib = phi_b; a.gamma = gamma_f; a.gammab = gamma_b;
    a.Cx = m->C[0]; a.Cy = m->C[1]; a.Cz = m->C[2]; a.bMagSf = m->bMagSf; a.bDelta = m->bDelta;
    a.rdt = rDeltaT; a.twoByk = 2.0 / std::max(k, 1e-15); a.lo = lo; a.hi = hi; a.scheme = scheme; a.wGiven = nullptr; a.nOffDiag = nf;
#define SE(NF) FFM_DISPATCH_W(m->A->maxW, LAUNCH_CELLS((k_scalar_eqns<W, NF>), mview(m), a))
    switch (nf) { case 1: SE(1); break; case 2: SE(2); break; case 3: SE(3); break; default: SE(4); break; }
#undef SE
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

// The common weights of a multivariateSelection scheme over nf fields (schemes[i]: 2 limitedLinear, 3 limitedLinear01; one k, one pair
// of bounds): equal, bit for bit, to ffm_fv_limited_limiter over the fields (running minimum) followed by ffm_fv_weights_from_limiter
extern "C" int ffm_fv_multivariate_weights(ffm_mesh *m, int nf, const int *schemes, double k, double lo, double hi, const double *phi_f,
                                           const double *const *vf, const double *const *gx, const double *const *gy, const double *const *gz,
                                           double *out_w)
{
    CHECK_M(m);
    if (nf < 1 || nf > MV_MAX || !schemes || !phi_f || !vf || !gx || !gy || !gz || !out_w) return FFM_ERR_ARG;
    MvWeights a;
    for (int i = 0; i < MV_MAX; i++) {
        const int q = i < nf ? i : 0;
        if (i < nf && ((schemes[q] != 2 && schemes[q] != 3) || !vf[q] || !gx[q] || !gy[q] || !gz[q])) return FFM_ERR_ARG;
        a.vf[i] = vf[q]; a.gx[i] = gx[q]; a.gy[i] = gy[q]; a.gz[i] = gz[q]; a.scheme[i] = schemes[q];
    }
    a.nf = nf; a.phi = phi_f; a.Cx = m->C[0]; a.Cy = m->C[1]; a.Cz = m->C[2]; a.twoByk = 2.0 / std::max(k, 1e-15); a.lo = lo; a.hi = hi; a.out = out_w;
    FFM_DISPATCH_W(m->A->maxW, LAUNCH_CELLS(k_mv_weights<W>, mview(m), a));
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

// ffm_fvm_scalar_transport_multi with the face weights given (a multivariate scheme's common weights) instead of one limiter per field:
// ddt(rho, vf_i) + div(phi, vf_i) [weights w_f] - laplacian(gamma, vf_i) == su_i ... for nf fields in one pass
extern "C" int ffm_fvm_scalar_transport_multi_w(ffm_mesh *m, int nf, const double *w_f, double rDeltaT, const double *rho, const double *rho0,
                                                const double *phi_f, const double *phi_b, const double *gamma_f, const double *gamma_b,
                                                const double *const *vf0, const double *const *f, const double *const *ref,
                                                const double *const *refGrad, const double *const *su, const double *const *su2,
                                                const double *const *sp, const double *const *expl3, double *const *diag,
                                                double *const *upper, double *const *lower, double *const *source)
{
    CHECK_M(m);
    if (nf < 1 || nf > FUSE_MAX || !w_f || !rho || !rho0 || !phi_f || !gamma_f || (m->B && (!phi_b || !gamma_b))) return FFM_ERR_ARG;
    if (!vf0 || !f || !ref || !refGrad || !diag || !upper || !lower || !source) return FFM_ERR_ARG;
    ScalarEqns a;
    for (int i = 0; i < FUSE_MAX; i++) {
        const int q = i < nf ? i : 0;
        if (i < nf && (!vf0[q] || !diag[q] || !source[q] || (m->B && (!f[q] || !ref[q] || !refGrad[q])))) return FFM_ERR_ARG;
        a.vf[i] = a.gx[i] = a.gy[i] = a.gz[i] = nullptr; a.vf0[i] = vf0[q];
        a.f[i] = f[q]; a.ref[i] = ref[q]; a.refGrad[i] = refGrad[q];
        a.su[i] = su ? su[q] : nullptr;
        a.su2[i] = su2 ? su2[q] : nullptr; a.sp[i] = sp ? sp[q] : nullptr;
        for (int e = 0; e < 3; e++) a.expl[i][e] = expl3 ? expl3[3 * q + e] : nullptr;
        if (a.expl[i][0] && (!a.expl[i][1] || !a.expl[i][2])) return FFM_ERR_ARG;
        a.diag[i] = diag[q]; a.upper[i] = upper[q]; a.lower[i] = lower[q]; a.src[i] = source[q];
    }
    a.rho = rho; a.rho0 = rho0; a.phi = phi_f; a.phib = phi_b; a.gamma = gamma_f; a.gammab = gamma_b;
    a.Cx = m->C[0]; a.Cy = m->C[1]; a.Cz = m->C[2]; a.bMagSf = m->bMagSf; a.bDelta = m->bDelta;
    a.rdt = rDeltaT; a.twoByk = 2.0; a.lo = 0.0; a.hi = 1.0; a.scheme = 0; a.wGiven = w_f;
    // off-diagonals: every field with non-null upper AND lower writes its own; the fields after the first null pair share what was written
    a.nOffDiag = 0;
    for (int i = 0; i < nf; i++) { if (upper[i] && lower[i]) { if (a.nOffDiag != i) return FFM_ERR_ARG; a.nOffDiag = i + 1; } else if ((upper[i] == nullptr) != (lower[i] == nullptr)) return FFM_ERR_ARG; }
#define SEW(NF) FFM_DISPATCH_W(m->A->maxW, LAUNCH_CELLS((k_scalar_eqns<W, NF, true>), mview(m), a))
    switch (nf) { case 1: SEW(1); break; case 2: SEW(2); break; case 3: SEW(3); break; default: SEW(4); break; }
#undef SEW
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

extern "C" int ffm_fvm_lust_source3(ffm_mesh *m, double rDeltaT, const double *phi_f, const double *rho0, const double *const *U0,
                                    const double *const *gx, const double *const *gy, const double *const *gz, double *const *source)
{
    CHECK_M(m);
    if (!phi_f || !rho0 || !U0 || !gx || !gy || !gz || !source) return FFM_ERR_ARG;
    if (!m->Cf[0]) { ffm_set_error("ffm_fvm_lust_source3: face centres not set (ffm_mesh_set_face_centres)"); return FFM_ERR_ARG; }
    LustSource a;
    for (int i = 0; i < 3; i++) {
        if (!U0[i] || !gx[i] || !gy[i] || !gz[i] || !source[i]) return FFM_ERR_ARG;
        a.gx[i] = gx[i]; a.gy[i] = gy[i]; a.gz[i] = gz[i]; a.U0[i] = U0[i]; a.src[i] = source[i];
    }
    a.phi = phi_f; a.rho0 = rho0; a.Cx = m->C[0]; a.Cy = m->C[1]; a.Cz = m->C[2]; a.Cfx = m->Cf[0]; a.Cfy = m->Cf[1]; a.Cfz = m->Cf[2];
    a.rdt = rDeltaT;
    FFM_DISPATCH_W(m->A->maxW, LAUNCH_CELLS(k_lust_source<W>, mview(m), a));
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

// ---- pressure-corrector passes (solver/pEqn.H:5-20) ---------------------------------------------------------------------------
// HbyA = rAU*UEqn.H() for the three components in one pass over the rows: the off-diagonal coefficients of the momentum matrix
// are read once instead of three times (k_matrix_H per component + one product kernel each); per component the arithmetic is
// that of k_matrix_H followed by the product, bit for bit.
struct HbyA3 {
    const double *upper, *lower, *src[3], *ic[3], *bc[3], *psi[3], *rAU;
    double *out[3];
};
template <int W>
__global__ __launch_bounds__(256) void k_matrix_HbyA3(MeshView q, HbyA3 a)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci;
        RowEnt<W> L, U; load_lower<W>(q.v, c, L); load_upper<W>(q.v, c, U);
        double al[W], au[W];
#pragma unroll
        for (int s = 0; s < W; s++) { al[s] = a.lower[L.f[s]]; au[s] = a.upper[U.f[s]]; }
        double bd[3] = {0.0, 0.0, 0.0}, bda = 0.0, bs[3] = {0.0, 0.0, 0.0};
        const int j = q.cellB[c];
        if (j >= 0) for (int t = q.bcStart[j]; t < q.bcStart[j + 1]; t++) {
            const int k = q.bcItem[t];
            const double i0 = a.ic[0][k], i1 = a.ic[1][k], i2 = a.ic[2][k];
            bd[0] += i0; bd[1] += i1; bd[2] += i2; bda += (i0 + i1 + i2) / 3.0;
            bs[0] += a.bc[0][k]; bs[1] += a.bc[1][k]; bs[2] += a.bc[2][k];
        }
        const double V = q.V[c], r = a.rAU ? a.rAU[c] : 1.0;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const double *psi = a.psi[d];
            double hl = 0.0;
#pragma unroll
            for (int s = 0; s < W; s++) if (L.on[s]) hl -= al[s] * psi[L.nb[s]];
#pragma unroll
            for (int s = 0; s < W; s++) if (U.on[s]) hl -= au[s] * psi[U.nb[s]];
            double h = (bda - bd[d]) * psi[c];
            h += hl + a.src[d][c];
            h += bs[d];
            h = h / V;
            a.out[d][c] = a.rAU ? r * h : h;
        }
    }
}

// fvMatrix<vector>::H() of the three components, times rAU when given (rAU null: H itself).  Arrays as ffm_fvm_H.
extern "C" int ffm_fvm_HbyA3(ffm_mesh *m, const double *upper, const double *lower, const double *const *source, const double *const *ic,
                             const double *const *bc, const double *const *psi, const double *rAU, double *const *out)
{
    CHECK_M(m);
    if (!upper || !lower || !source || !ic || !bc || !psi || !out) return FFM_ERR_ARG;
    HbyA3 a; a.upper = upper; a.lower = lower; a.rAU = rAU;
    for (int d = 0; d < 3; d++) {
        if (!source[d] || !psi[d] || !out[d] || (m->B && (!ic[d] || !bc[d]))) return FFM_ERR_ARG;
        a.src[d] = source[d]; a.ic[d] = ic[d]; a.bc[d] = bc[d]; a.psi[d] = psi[d]; a.out[d] = out[d];
    }
    FFM_DISPATCH_W(m->A->maxW, LAUNCH_CELLS(k_matrix_HbyA3<W>, mview(m), a));
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

// fvc::flux(rho*v) on the internal faces without the product fields: (w (rho v)_P + (1 - w) (rho v)_N) . Sf, the products formed
// where they are used (the values k_flux reads from the stored product fields)
__global__ __launch_bounds__(256) void k_flux_rho(MeshView q, const double *__restrict__ rho, const double *__restrict__ vx,
                                                  const double *__restrict__ vy, const double *__restrict__ vz, double *__restrict__ out)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci; const double rP = rho[c]; const double Px = rP * vx[c], Py = rP * vy[c], Pz = rP * vz[c];
        FOR_OWN_FACES(q, c, e, nb) {
            const double w = q.w[e], rN = rho[nb];
            out[e] = (w * Px + (1.0 - w) * (rN * vx[nb])) * q.Sfx[e] + (w * Py + (1.0 - w) * (rN * vy[nb])) * q.Sfy[e] + (w * Pz + (1.0 - w) * (rN * vz[nb])) * q.Sfz[e];
        }
    }
}
extern "C" int ffm_fvc_flux_rho(ffm_mesh *m, const double *rho, const double *vx, const double *vy, const double *vz, double *out_f)
{
    CHECK_M(m);
    if (!rho || !vx || !vy || !vz || !out_f) return FFM_ERR_ARG;
    LAUNCH_CELLS(k_flux_rho, mview(m), rho, vx, vy, vz, out_f);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

// ---- three face passes of the pressure corrector (solver/pEqn.H:9-19,43-44), each the fusion of two per-operator passes with their
// arithmetic (fvc::snGrad's delta*(N - P); fvMatrix::flux's upper*psi_N - lower*psi_P), so that a face field is not written and read back:
//   ffm_pc_phig      phig = -rhorAUf*ghf*fvc::snGrad(rho)*magSf
//   ffm_pc_phiHbyA   phiHbyA = (fvc::flux(rho*HbyA) + rhorAUf*ddtCorr) + phig            (k_flux_rho + the two additions)
//   ffm_pc_flux      fl = p_rghEqn.flux();  phi = phiHbyA + fl;  t = (fl + phig)/rhorAUf   (the argument of fvc::reconstruct)
// every slot of the cell's slice row, padding included (nb < 0)
#define PC_ALL_SLOTS(q, c, e, nb)                                                      \
    const int sl_ = (c) >> 6, lane_ = (c)&63;                                          \
    const int ub_ = up_base((q).v, sl_), uw_ = up_width((q).v, sl_);                   \
    for (int s_ = 0, e = ub_ + lane_, nb = 0; s_ < uw_ && ((nb = (q).v.upNbr[e]), true); s_++, e += 64)
__global__ __launch_bounds__(256) void k_pc_phig(MeshView q, const double *__restrict__ rhorAUf, const double *__restrict__ ghf,
                                                 const double *__restrict__ rho, double *__restrict__ out)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci; const double P = rho[c];
        PC_ALL_SLOTS(q, c, e, nb) {
            if (nb < 0) { out[e] = 0.0; continue; }              // padding entries of the native face layout hold 0
            const double sg = q.delta[e] * (rho[nb] - P); out[e] = -rhorAUf[e] * ghf[e] * sg * q.magSf[e];
        }
    }
}
__global__ __launch_bounds__(256) void k_pc_phiHbyA(MeshView q, const double *__restrict__ rho, const double *__restrict__ vx,
                                                    const double *__restrict__ vy, const double *__restrict__ vz, const double *__restrict__ rhorAUf,
                                                    const double *__restrict__ dc, const double *__restrict__ phig, double *__restrict__ out)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci; const double rP = rho[c]; const double Px = rP * vx[c], Py = rP * vy[c], Pz = rP * vz[c];
        PC_ALL_SLOTS(q, c, e, nb) {
            if (nb < 0) { out[e] = 0.0; continue; }
            const double w = q.w[e], rN = rho[nb];
            const double fl = (w * Px + (1.0 - w) * (rN * vx[nb])) * q.Sfx[e] + (w * Py + (1.0 - w) * (rN * vy[nb])) * q.Sfy[e] + (w * Pz + (1.0 - w) * (rN * vz[nb])) * q.Sfz[e];
            out[e] = (fl + rhorAUf[e] * dc[e]) + phig[e];
        }
    }
}
__global__ __launch_bounds__(256) void k_pc_flux(MeshView q, const double *__restrict__ upper, const double *__restrict__ lower,
                                                 const double *__restrict__ psi, const double *__restrict__ phiHbyA, const double *__restrict__ phig,
                                                 const double *__restrict__ rhorAUf, double *__restrict__ fl, double *__restrict__ phi,
                                                 double *__restrict__ t)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci; const double P = psi[c];
        PC_ALL_SLOTS(q, c, e, nb) {
            if (nb < 0) { fl[e] = 0.0; phi[e] = 0.0; t[e] = 0.0; continue; }
            const double f = upper[e] * psi[nb] - lower[e] * P;
            fl[e] = f; phi[e] = phiHbyA[e] + f;
            t[e] = rhorAUf[e] != 0.0 ? (f + phig[e]) / rhorAUf[e] : 0.0;
        }
    }
}
// the face flux fvc::reconstruct takes in solver/UEqn.H:23-29: t = (-ghf*fvc::snGrad(rho) - fvc::snGrad(p_rgh))*magSf (two snGrad passes and
// their combination in one)
__global__ __launch_bounds__(256) void k_ue_buoyancy_flux(MeshView q, const double *__restrict__ ghf, const double *__restrict__ rho,
                                                         const double *__restrict__ prgh, double *__restrict__ out)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci; const double rP = rho[c], pP = prgh[c];
        PC_ALL_SLOTS(q, c, e, nb) {
            if (nb < 0) { out[e] = 0.0; continue; }
            const double sgr = q.delta[e] * (rho[nb] - rP), sgp = q.delta[e] * (prgh[nb] - pP);
            out[e] = (-ghf[e] * sgr - sgp) * q.magSf[e];
        }
    }
}
extern "C" int ffm_ue_buoyancy_flux(ffm_mesh *m, const double *ghf, const double *rho, const double *p_rgh, double *t_f)
{
    CHECK_M(m);
    if (!ghf || !rho || !p_rgh || !t_f) return FFM_ERR_ARG;
    LAUNCH_CELLS(k_ue_buoyancy_flux, mview(m), ghf, rho, p_rgh, t_f);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}
extern "C" int ffm_pc_phig(ffm_mesh *m, const double *rhorAUf, const double *ghf, const double *rho, double *phig)
{
    CHECK_M(m);
    if (!rhorAUf || !ghf || !rho || !phig) return FFM_ERR_ARG;
    LAUNCH_CELLS(k_pc_phig, mview(m), rhorAUf, ghf, rho, phig);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}
extern "C" int ffm_pc_phiHbyA(ffm_mesh *m, const double *rho, const double *vx, const double *vy, const double *vz, const double *rhorAUf,
                              const double *ddtCorr, const double *phig, double *phiHbyA)
{
    CHECK_M(m);
    if (!rho || !vx || !vy || !vz || !rhorAUf || !ddtCorr || !phig || !phiHbyA) return FFM_ERR_ARG;
    LAUNCH_CELLS(k_pc_phiHbyA, mview(m), rho, vx, vy, vz, rhorAUf, ddtCorr, phig, phiHbyA);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}
extern "C" int ffm_pc_flux(ffm_mesh *m, const double *upper, const double *lower, const double *psi, const double *phiHbyA, const double *phig,
                           const double *rhorAUf, double *flux_f, double *phi_f, double *t_f)
{
    CHECK_M(m);
    if (!upper || !lower || !psi || !phiHbyA || !phig || !rhorAUf || !flux_f || !phi_f || !t_f) return FFM_ERR_ARG;
    LAUNCH_CELLS(k_pc_flux, mview(m), upper, lower, psi, phiHbyA, phig, rhorAUf, flux_f, phi_f, t_f);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

// p_rghEqn of solver/pEqn.H:28-36 in one pass over the rows:
//     fvm::ddt(psi, p_rgh) + fvc::ddt(psi, rho)*gh + fvc::ddt(psi)*pRef + fvc::div(phiHbyA) - fvm::laplacian(rhorAUf, p_rgh)
// i.e. the laplacian coefficients (k_fvm_transport), fvc::div(phiHbyA) (k_face_sum), the three explicit terms (one source
// update each, in the reference's order) and addBoundaryDiag / addBoundarySource (k_add_boundary), with their arithmetic.
struct PEqn {
    const double *psi, *psi0, *p0, *rho, *rho0, *gh, *gamma, *phiHbyA, *phiHbyAb, *ic, *bc;
    double *upper, *lower, *diag, *src;
    double rdt, pRef;
};
template <int W>
__global__ __launch_bounds__(256) void k_p_rgh_eqn(MeshView q, PEqn a)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci;
        RowEnt<W> L, U; load_lower<W>(q.v, c, L); load_upper<W>(q.v, c, U);
        double dLap = 0.0, acc = 0.0;
#pragma unroll
        for (int s = 0; s < W; s++) if (L.on[s]) { const int e = L.f[s]; dLap -= a.gamma[e] * q.magSf[e] * q.delta[e]; }
#pragma unroll
        for (int s = 0; s < W; s++) if (U.on[s]) {
            const int e = U.f[s];
            const double g = a.gamma[e] * q.magSf[e] * q.delta[e];
            dLap -= g;
            a.upper[e] = -g; a.lower[e] = -g;
        }
#pragma unroll
        for (int s = 0; s < W; s++) if (L.on[s]) acc = acc - a.phiHbyA[L.f[s]];
#pragma unroll
        for (int s = 0; s < W; s++) if (U.on[s]) acc += a.phiHbyA[U.f[s]];
        const double V = q.V[c];
        double d = a.rdt * a.psi[c] * V;
        d = d - dLap;
        double bd = 0.0, bs = 0.0;      // (the boundary sums are formed in k_add_boundary's order: onto the finished diag / source)
        const int j = q.cellB[c];
        if (j >= 0) for (int t = q.bcStart[j]; t < q.bcStart[j + 1]; t++) acc += a.phiHbyAb[q.bcItem[t]];
        const double div = acc / V;
        double s = ((a.rdt * a.psi0[c] * a.p0[c] * V - V * (a.rdt * (a.psi[c] * a.rho[c] - a.psi0[c] * a.rho0[c]) * a.gh[c]))
                    - V * (a.rdt * (a.psi[c] - a.psi0[c]) * a.pRef)) - V * div;
        if (j >= 0) for (int t = q.bcStart[j]; t < q.bcStart[j + 1]; t++) { const int k = q.bcItem[t]; d += a.ic[k]; s += a.bc[k]; }
        (void)bd; (void)bs;
        a.diag[c] = d; a.src[c] = s;
    }
}
extern "C" int ffm_fvm_pressure_eqn(ffm_mesh *m, double rDeltaT, const double *psi, const double *psi0, const double *p0, const double *rho,
                                    const double *rho0, const double *gh, double pRef, const double *gamma_f, const double *phiHbyA_f,
                                    const double *phiHbyA_b, const double *ic, const double *bc, double *upper, double *lower, double *diagOut,
                                    double *sourceOut)
{
    CHECK_M(m);
    if (!psi || !psi0 || !p0 || !rho || !rho0 || !gh || !gamma_f || !phiHbyA_f || !upper || !lower || !diagOut || !sourceOut ||
        (m->B && (!phiHbyA_b || !ic || !bc))) return FFM_ERR_ARG;
    PEqn a{psi, psi0, p0, rho, rho0, gh, gamma_f, phiHbyA_f, phiHbyA_b, ic, bc, upper, lower, diagOut, sourceOut, rDeltaT, pRef};
    FFM_DISPATCH_W(m->A->maxW, LAUNCH_CELLS(k_p_rgh_eqn<W>, mview(m), a));
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

// ---- LES momentum / k-equation terms built on fvc::grad(U) (N2: kEqn of cases/steckler/constant/turbulenceProperties:18-30) ------
// turbulence->divDevRhoReff(U) (solver/UEqn.H:12) = - fvc::div((rho*nuEff)*dev2(T(fvc::grad(U)))) - fvm::laplacian(rho*nuEff, U):
// the explicit part  div(X), X = gamma*dev2(T(gradU)), Gauss linear, in one pass over the rows from the nine cell gradients
// (g[i][j] = d_i U_j, e.g. of ffm_fvc_grad_multi) -- face tensors formed on the fly; on patch faces gaussGrad's corrected
// boundary gradient gb = gc + n (x) (snGrad(U) - n.gc), snGrad = deltaCoeffs (U_b - U_c), times gamma_b.
struct DivDev2T {
    const double *g[3][3];          // [i][j]
    const double *gam, *gamb, *U[3], *Ub[3];
    const double *bMagSf, *bDelta; const int *bCells;
    double *out[3];
};
__device__ __forceinline__ void dev2T_of(const double g[3][3], double s, double X[3][3])
{
    // t = T(g); dev2(t) = t - (2/3) tr(t) I; X = s*dev2(t)
    const double tr = (g[0][0] + g[1][1]) + g[2][2];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) { double t = g[j][i]; if (i == j) t = t - (2.0 / 3.0) * tr; X[i][j] = s * t; }
}
template <int W>
__global__ __launch_bounds__(256) void k_div_dev2T(MeshView q, DivDev2T a)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci;
        RowEnt<W> L, U; load_lower<W>(q.v, c, L); load_upper<W>(q.v, c, U);
        double gP[3][3], XP[3][3];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) gP[i][j] = a.g[i][j][c];
        dev2T_of(gP, a.gam[c], XP);
        double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int pass = 0; pass < 2; pass++)
#pragma unroll
            for (int s = 0; s < W; s++) {
                const bool on = pass == 0 ? L.on[s] : U.on[s];
                if (!on) continue;
                const int nb = pass == 0 ? L.nb[s] : U.nb[s], e = pass == 0 ? L.f[s] : U.f[s];
                double gN[3][3], XN[3][3];
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                    for (int j = 0; j < 3; j++) gN[i][j] = a.g[i][j][nb];
                dev2T_of(gN, a.gam[nb], XN);
                const double w = q.w[e], S[3] = {q.Sfx[e], q.Sfy[e], q.Sfz[e]};
                // owner of the face: the neighbour cell for a lower face (weight on the owner)
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    double fl = 0.0;
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        const double Xf = pass == 0 ? w * XN[i][j] + (1.0 - w) * XP[i][j] : w * XP[i][j] + (1.0 - w) * XN[i][j];
                        fl += S[i] * Xf;
                    }
                    acc[j] = pass == 0 ? acc[j] - fl : acc[j] + fl;
                }
            }
        const int jb = q.cellB[c];
        if (jb >= 0) for (int t = q.bcStart[jb]; t < q.bcStart[jb + 1]; t++) {
            const int k = q.bcItem[t];
            const double S[3] = {q.bSfx[k], q.bSfy[k], q.bSfz[k]}, mag = a.bMagSf[k];
            const double n[3] = {S[0] / mag, S[1] / mag, S[2] / mag};
            double gb[3][3], Xb[3][3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const double sn = a.bDelta[k] * (a.Ub[j][k] - a.U[j][c]);
                const double ng = (n[0] * gP[0][j] + n[1] * gP[1][j]) + n[2] * gP[2][j];
#pragma unroll
                for (int i = 0; i < 3; i++) gb[i][j] = gP[i][j] + n[i] * (sn - ng);
            }
            dev2T_of(gb, a.gamb[k], Xb);
#pragma unroll
            for (int j = 0; j < 3; j++) acc[j] += (S[0] * Xb[0][j] + S[1] * Xb[1][j]) + S[2] * Xb[2][j];
        }
        const double V = q.V[c];
#pragma unroll
        for (int j = 0; j < 3; j++) a.out[j][c] = acc[j] / V;
    }
}
// g: nine device pointers, g[3*i + j] = d_i U_j of the cells; out[j]: component j of fvc::div(gamma*dev2(T(grad(U))))
extern "C" int ffm_fvc_div_dev2T_gradU(ffm_mesh *m, const double *const *g, const double *gamma, const double *gamma_b, const double *const *U,
                                       const double *const *U_b, double *const *out)
{
    CHECK_M(m);
    if (!g || !gamma || !U || !out || (m->B && (!gamma_b || !U_b))) return FFM_ERR_ARG;
    DivDev2T a;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) { if (!g[3 * i + j]) return FFM_ERR_ARG; a.g[i][j] = g[3 * i + j]; }
        if (!U[i] || !out[i] || (m->B && !U_b[i])) return FFM_ERR_ARG;
        a.U[i] = U[i]; a.Ub[i] = U_b ? U_b[i] : nullptr; a.out[i] = out[i];
    }
    a.gam = gamma; a.gamb = gamma_b; a.bMagSf = m->bMagSf; a.bDelta = m->bDelta; a.bCells = m->bCells;
    FFM_DISPATCH_W(m->A->maxW, LAUNCH_CELLS(k_div_dev2T<W>, mview(m), a));
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

// kEqn::correct (LESModels/kEqn/kEqn.C): G = nut*(gradU && dev(twoSymm(gradU))), per cell
__global__ void k_les_G(long n, const double *g00, const double *g01, const double *g02, const double *g10, const double *g11, const double *g12,
                        const double *g20, const double *g21, const double *g22, const double *__restrict__ nut, double *__restrict__ G)
{
    GRID_STRIDE(i, n) {
        const double g[3][3] = {{g00[i], g01[i], g02[i]}, {g10[i], g11[i], g12[i]}, {g20[i], g21[i], g22[i]}};
        double ts[3][3];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) ts[a][b] = g[a][b] + g[b][a];
        const double tr = (ts[0][0] + ts[1][1]) + ts[2][2];
#pragma unroll
        for (int a = 0; a < 3; a++) ts[a][a] = ts[a][a] - (1.0 / 3.0) * tr;
        double s = 0.0;
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) s += g[a][b] * ts[a][b];
        G[i] = nut[i] * s;
    }
}
extern "C" int ffm_les_keqn_G(ffm_mesh *m, const double *const *g, const double *nut, double *G)
{
    CHECK_M(m);
    if (!g || !nut || !G) return FFM_ERR_ARG;
    for (int k = 0; k < 9; k++) if (!g[k]) return FFM_ERR_ARG;
    LAUNCH(k_les_G, m->N, (long)m->N, g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], nut, G);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}
