// ffm_fv.hip -- finite-volume assembly (fvm::) and explicit operators (fvc::) on the device mesh.
//
// Replaces (OpenFOAM-dev @940e28f, not vendored in the reference):
//   src/finiteVolume/finiteVolume/fvc/{fvcSurfaceIntegrate,fvcSnGrad,fvcGrad,fvcReconstruct,fvcFlux}.C
//   src/finiteVolume/finiteVolume/convectionSchemes/gaussConvectionScheme/gaussConvectionScheme.C
//   src/finiteVolume/finiteVolume/laplacianSchemes/gaussLaplacianScheme/gaussLaplacianScheme.C
//   src/finiteVolume/finiteVolume/ddtSchemes/EulerDdtScheme/EulerDdtScheme.C
//   src/finiteVolume/interpolation/surfaceInterpolation/limitedSchemes/{limitedLinear,LimitedScheme,NVDTVD}.H
//   src/finiteVolume/fvMatrices/fvMatrix/fvMatrix.C (negSumDiag, addBoundaryDiag/Source, A, H, flux)
//   src/finiteVolume/fields/fvPatchFields/basic/mixed/mixedFvPatchField.C (coefficient formulas)
// reached from the reference at solver/UEqn.H:3-30, solver/YEEqn.H:43-111, solver/pEqn.H:3-60,
// solver/rhoEqn.H:33-43, solver/phrghEqn.H:25-56; schemes from cases/steckler/system/fvSchemes:18-76.
//
// Layout: cell fields [N] in the library's cell order; face fields [nNative] in the sliced
// owner-ELL face layout of the LDU (owner implicit, padding entries hold 0); boundary fields [B],
// patches concatenated.  Every cell-sum kernel is one thread per row: it first adds the faces
// whose neighbour is the cell (lower entries), then the faces the cell owns (upper slots), then
// its boundary faces in (patch, face) order -- the order of the reference's face loops -- so the
// sums carry no atomics and are reproducible.  All kernels are HBM-bound streaming/gather kernels.
#include "ffm_internal.hpp"
#include "ffm_device.hpp"
#include <algorithm>
#include <cmath>

#include "ffm_mesh.hpp"

template <class T> static int up(ffm_ctx *c, T **d, const std::vector<T> &v)
{
    return ffm_upload_vec(c, d, v);
}

extern "C" int ffm_mesh_create(ffm_ldu *A, const double *V, const double *C, const double *Sf, const double *magSf,
                               const double *weights, const double *deltaCoeffs, int nPatches, const int *patchSizes,
                               const int *const *faceCells, const double *const *pSf, const double *const *pDelta,
                               ffm_mesh **out)
{
    if (!A || !V || !C || !out || nPatches < 0) return FFM_ERR_ARG;
    if (!A->identity) { ffm_set_error("ffm_mesh_create needs a mesh numbered with ffm_renumber_levels (native order)"); return FFM_ERR_UNSUPPORTED; }
    ffm_mesh *m = new ffm_mesh();
    m->A = A; m->ctx = A->ctx; m->N = A->nCells; m->F = A->nFaces; m->nNat = A->upTotal; m->nPatches = nPatches;
    const int N = m->N, F = m->F, nNat = m->nNat;
    m->patchOff.assign(nPatches + 1, 0);
    for (int p = 0; p < nPatches; p++) m->patchOff[p + 1] = m->patchOff[p] + patchSizes[p];
    m->B = m->patchOff[nPatches];
    const int B = m->B;
    // face geometry: caller (LDU) face order -> native
    // (one staging array for all six face fields: filled and scattered by the host threads, then uploaded)
    std::vector<double> stage;
    stage.reserve(std::max(nNat, 1));
    int rc = FFM_OK;
    auto upNative = [&](double **dst, const double *src, double fill) -> int {
        if (stage.empty()) stage.resize(std::max(nNat, 1));
        const int *c2n = A->h_callerToNative.data(); double *vp = stage.data();
        ffm_parallel_for(nNat, [=](long lo, long hi) { for (long q = lo; q < hi; q++) vp[q] = fill; });
        ffm_parallel_for(F, [=](long lo, long hi) { for (long f = lo; f < hi; f++) vp[c2n[f]] = src[f]; });      // (distinct destinations)
        return up(m->ctx, dst, stage);
    };
    auto upCells = [&](double **dst, const double *src) -> int {          // straight from the caller's array
        FFM_HIP(hipMalloc((void **)dst, sizeof(double) * std::max<size_t>(N, 1)));
        return ffm_h2d(m->ctx, *dst, src, sizeof(double) * (size_t)N);
    };
    if ((rc = upCells(&m->V, V))) return rc;
    for (int d = 0; d < 3; d++) {
        if ((rc = upCells(&m->C[d], C + (size_t)d * N))) return rc;
        if ((rc = upNative(&m->Sf[d], Sf + (size_t)d * F, 0.0))) return rc;
    }
    if ((rc = upNative(&m->magSf, magSf, 0.0))) return rc;
    if ((rc = upNative(&m->delta, deltaCoeffs, 0.0))) return rc;
    if ((rc = upNative(&m->w, weights, 0.5))) return rc;
    std::vector<double>().swap(stage);
    // boundary
    std::vector<int> bc(std::max(B, 1), 0);
    std::vector<double> bS[3], bM(std::max(B, 1), 0.0), bD(std::max(B, 1), 0.0);
    for (int d = 0; d < 3; d++) bS[d].assign(std::max(B, 1), 0.0);
    for (int p = 0; p < nPatches; p++) for (int i = 0; i < patchSizes[p]; i++) {
        const int k = m->patchOff[p] + i;
        bc[k] = faceCells[p][i];
        double s2 = 0;
        for (int d = 0; d < 3; d++) { bS[d][k] = pSf[p][(size_t)d * patchSizes[p] + i]; s2 += bS[d][k] * bS[d][k]; }
        bM[k] = std::sqrt(s2); bD[k] = pDelta[p][i];
        if (bc[k] < 0 || bc[k] >= N) { ffm_set_error("patch %d: faceCell out of range", p); return FFM_ERR_ARG; }
    }
    if ((rc = up(m->ctx, &m->bCells, bc))) return rc;
    for (int d = 0; d < 3; d++) if ((rc = up(m->ctx, &m->bSf[d], bS[d]))) return rc;
    if ((rc = up(m->ctx, &m->bMagSf, bM))) return rc;
    if ((rc = up(m->ctx, &m->bDelta, bD))) return rc;
    // boundary-cell CSR, items in (patch, face) order
    std::vector<int> order(B);
    for (int k = 0; k < B; k++) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return bc[a] < bc[b]; });
    std::vector<int> cellB(N, -1), start;
    for (int j = 0; j < B; j++) if (j == 0 || bc[order[j]] != bc[order[j - 1]]) { cellB[bc[order[j]]] = (int)start.size(); start.push_back(j); }
    m->nBC = (int)start.size(); start.push_back(B);
    if ((rc = up(m->ctx, &m->cellB, cellB))) return rc;
    if ((rc = up(m->ctx, &m->bcStart, start))) return rc;
    if ((rc = up(m->ctx, &m->bcItem, order))) return rc;
    // inverse reconstruction tensor per cell (host, once): T = sum_f Sf (x) Sf / magSf over all faces of the cell
    std::vector<double> T((size_t)6 * N, 0.0);   // xx xy xz yy yz zz
    auto addT = [&](int c, const double s[3], double mag) {
        if (mag <= 0) return;
        T[0 * (size_t)N + c] += s[0] * s[0] / mag; T[1 * (size_t)N + c] += s[0] * s[1] / mag; T[2 * (size_t)N + c] += s[0] * s[2] / mag;
        T[3 * (size_t)N + c] += s[1] * s[1] / mag; T[4 * (size_t)N + c] += s[1] * s[2] / mag; T[5 * (size_t)N + c] += s[2] * s[2] / mag;
    };
    {
        // owner/neighbour of caller faces: recover from the native layout on the host
        std::vector<int> nat2caller(nNat, -1);
        for (int f = 0; f < F; f++) nat2caller[A->h_callerToNative[f]] = f;
        std::vector<int> hUpNbr(std::max(nNat, 1));
        FFM_HIP(hipStreamSynchronize(A->ctx->stream));              // (the tables were uploaded on the context's non-blocking stream)
        FFM_TRY(ffm_d2h(A->ctx, hUpNbr.data(), A->upNbr, sizeof(int) * nNat));
        std::vector<int> hUpOff(A->nSlices + 1);
        FFM_TRY(ffm_d2h(A->ctx, hUpOff.data(), A->upOff, sizeof(int) * (A->nSlices + 1)));
        for (int sl = 0; sl < A->nSlices; sl++) {
            const int wdt = (hUpOff[sl + 1] - hUpOff[sl]) / 64;
            for (int s = 0; s < wdt; s++) for (int lane = 0; lane < 64; lane++) {
                const int e = hUpOff[sl] + s * 64 + lane, c = sl * 64 + lane;
                if (c >= N || hUpNbr[e] < 0) continue;
                const int f = nat2caller[e];
                const double sv[3] = {Sf[f], Sf[(size_t)F + f], Sf[(size_t)2 * F + f]};
                addT(c, sv, magSf[f]); addT(hUpNbr[e], sv, magSf[f]);
            }
        }
        for (int k = 0; k < B; k++) { const double sv[3] = {bS[0][k], bS[1][k], bS[2][k]}; addT(bc[k], sv, bM[k]); }
        std::vector<double> inv((size_t)6 * N, 0.0);
        // 2-D / 1-D meshes (empty patches): the tensor has no entries in the missing directions.  As OpenFOAM's inv(tensorField)
        // does, the missing directions are found from the first cell (diagonal entry negligible against the tensor's magnitude),
        // 1 is added on those diagonals before the inversion and taken off again afterwards
        bool rm[3] = {false, false, false};
        if (N > 0) {
            const double t0[6] = {T[0], T[(size_t)N], T[(size_t)2 * N], T[(size_t)3 * N], T[(size_t)4 * N], T[(size_t)5 * N]};
            const double scale = t0[0] * t0[0] + 2 * t0[1] * t0[1] + 2 * t0[2] * t0[2] + t0[3] * t0[3] + 2 * t0[4] * t0[4] + t0[5] * t0[5];
            if (scale > 0) { rm[0] = t0[0] * t0[0] / scale < 1e-15; rm[1] = t0[3] * t0[3] / scale < 1e-15; rm[2] = t0[5] * t0[5] / scale < 1e-15; }
        }
        const double *Tp = T.data(); double *invp = inv.data();
        const bool rm0 = rm[0], rm1 = rm[1], rm2 = rm[2];
        ffm_parallel_for(N, [=](long c0_, long c1_) { const double *T = Tp; double *inv = invp; const bool rm[3] = {rm0, rm1, rm2};
        for (long c = c0_; c < c1_; c++) {
            const double a = T[c] + (rm[0] ? 1.0 : 0.0), b = T[(size_t)N + c], cc = T[(size_t)2 * N + c], d = T[(size_t)3 * N + c] + (rm[1] ? 1.0 : 0.0),
                         e = T[(size_t)4 * N + c], f = T[(size_t)5 * N + c] + (rm[2] ? 1.0 : 0.0);
            const double det = a * (d * f - e * e) - b * (b * f - e * cc) + cc * (b * e - d * cc);
            if (det == 0) continue;     // a cell without faces (ghost cells): reconstruct returns 0 there
            inv[c] = (d * f - e * e) / det - (rm[0] ? 1.0 : 0.0); inv[(size_t)N + c] = (cc * e - b * f) / det; inv[(size_t)2 * N + c] = (b * e - cc * d) / det;
            inv[(size_t)3 * N + c] = (a * f - cc * cc) / det - (rm[1] ? 1.0 : 0.0); inv[(size_t)4 * N + c] = (b * cc - a * e) / det;
            inv[(size_t)5 * N + c] = (a * d - b * b) / det - (rm[2] ? 1.0 : 0.0);
        } });
        if ((rc = up(m->ctx, &m->invT, inv))) return rc;
    }
    *out = m;
    return FFM_OK;
}

extern "C" int ffm_mesh_destroy(ffm_mesh *m)
{
    if (!m) return FFM_OK;
    hipStreamSynchronize(m->ctx->stream);
    for (int d_ = 0; d_ < 3; d_++) hipFree(m->corr[d_]);
    hipFree(m->V); hipFree(m->magSf); hipFree(m->delta); hipFree(m->w); hipFree(m->invT); hipFree(m->bCells);
    hipFree(m->bMagSf); hipFree(m->bDelta); hipFree(m->cellB); hipFree(m->bcStart); hipFree(m->bcItem);
    for (int d = 0; d < 3; d++) { hipFree(m->C[d]); hipFree(m->Sf[d]); hipFree(m->bSf[d]); hipFree(m->Cf[d]); }
    delete m;
    return FFM_OK;
}
extern "C" int ffm_mesh_nboundary(const ffm_mesh *m) { return m ? m->B : FFM_ERR_ARG; }
extern "C" int ffm_mesh_nnative(const ffm_mesh *m) { return m ? m->nNat : FFM_ERR_ARG; }

// caller (LDU) face order <-> native face layout, host arrays (tests, I/O)
extern "C" int ffm_faces_to_native(const ffm_mesh *m, const double *lduOrder, double *native_d)
{
    if (!m || !lduOrder || !native_d) return FFM_ERR_ARG;
    std::vector<double> v(std::max(m->nNat, 1), 0.0);
    for (int f = 0; f < m->F; f++) v[m->A->h_callerToNative[f]] = lduOrder[f];
    // on the context's stream, not the null stream: the destination is usually a block fresh from ffm_malloc, whose zero-fill is
    // still queued on that (non-blocking) stream -- a null-stream copy can land before it and be wiped (seen with two
    // processes sharing the GPU: flux fields arriving as zeros, round 2)
    FFM_HIP(hipMemcpyAsync(native_d, v.data(), sizeof(double) * m->nNat, hipMemcpyHostToDevice, m->ctx->stream));
    FFM_HIP(hipStreamSynchronize(m->ctx->stream));
    return FFM_OK;
}
extern "C" int ffm_faces_from_native(const ffm_mesh *m, const double *native_d, double *lduOrder)
{
    if (!m || !lduOrder || !native_d) return FFM_ERR_ARG;
    std::vector<double> v(std::max(m->nNat, 1));
    FFM_HIP(hipMemcpyAsync(v.data(), native_d, sizeof(double) * m->nNat, hipMemcpyDeviceToHost, m->ctx->stream));
    FFM_HIP(hipStreamSynchronize(m->ctx->stream));
    for (int f = 0; f < m->F; f++) lduOrder[f] = v[m->A->h_callerToNative[f]];
    return FFM_OK;
}

// fvc::interpolate with given weights (NULL: the mesh's linear weights)
__global__ void k_interpolate(MeshView q, const double *__restrict__ wf, const double *__restrict__ vf, double *__restrict__ out)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci; const double P = vf[c];
        FOR_OWN_FACES(q, c, e, nb) { const double w = wf ? wf[e] : q.w[e]; out[e] = w * P + (1.0 - w) * vf[nb]; }
    }
}
// fvc::snGrad (uncorrected)
__global__ void k_snGrad(MeshView q, const double *__restrict__ vf, double *__restrict__ out)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci; const double P = vf[c];
        FOR_OWN_FACES(q, c, e, nb) out[e] = q.delta[e] * (vf[nb] - P);
    }
}
// correctedSnGrad<Type>::correction(vf) for one scalar component: nonOrthCorrectionVectors & linear-interpolate(grad(vf))
__global__ void k_snGrad_correction(MeshView q, const double *__restrict__ cx, const double *__restrict__ cy, const double *__restrict__ cz,
                                    const double *__restrict__ gx, const double *__restrict__ gy, const double *__restrict__ gz,
                                    double *__restrict__ out)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci; const double Px = gx[c], Py = gy[c], Pz = gz[c];
        FOR_OWN_FACES(q, c, e, nb) {
            const double w = q.w[e];
            out[e] = (cx[e] * (w * Px + (1.0 - w) * gx[nb]) + cy[e] * (w * Py + (1.0 - w) * gy[nb])) + cz[e] * (w * Pz + (1.0 - w) * gz[nb]);
        }
    }
}
// fvc::flux(v) = linear-interpolate(v) & Sf for a vector field given as three component arrays
__global__ void k_flux(MeshView q, const double *__restrict__ vx, const double *__restrict__ vy, const double *__restrict__ vz,
                       double *__restrict__ out)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci; const double Px = vx[c], Py = vy[c], Pz = vz[c];
        FOR_OWN_FACES(q, c, e, nb) {
            const double w = q.w[e];
            out[e] = (w * Px + (1.0 - w) * vx[nb]) * q.Sfx[e] + (w * Py + (1.0 - w) * vy[nb]) * q.Sfy[e] + (w * Pz + (1.0 - w) * vz[nb]) * q.Sfz[e];
        }
    }
}
// boundary snGrad: delta_b*(vb - vf[faceCell])
__global__ void k_snGrad_b(int B, const int *__restrict__ fc, const double *__restrict__ delta, const double *__restrict__ vf,
                           const double *__restrict__ vb, double *__restrict__ out)
{ GRID_STRIDE(k, B) out[k] = delta[k] * (vb[k] - vf[fc[k]]); }

// limited-scheme face weights (NVDTVD::r, limitedLinearLimiter, LimitedLimiter, weights())
// scheme: 0 upwind, 1 linear, 2 limitedLinear, 3 limitedLinear01
// MODE 0: the weights; MODE 1: the limiter itself (limitedSurfaceInterpolationScheme::limiter; schemes 2 and 3); MODE 2: out = min(out,
// limiter) -- the running minimum over the fields of a multivariateSelection scheme (ffm_fv_limited_limiter)
template <int MODE>
__global__ void k_limited_weights(MeshView q, int scheme, double twoByk, double lo, double hi, const double *__restrict__ phi,
                                  const double *__restrict__ vf, const double *__restrict__ gx, const double *__restrict__ gy,
                                  const double *__restrict__ gz, const double *__restrict__ Cx, const double *__restrict__ Cy,
                                  const double *__restrict__ Cz, double *__restrict__ out)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci;
        const double P = vf ? vf[c] : 0.0;
        FOR_OWN_FACES(q, c, e, nb) {
            const double flux = phi[e];
            const double p0 = flux >= 0 ? 1.0 : 0.0;
            double wgt;
            if (scheme == 0 || scheme == 5) wgt = p0;                       // upwind; linearUpwind<Type>::weights = upwind's
            else if (scheme == 1) wgt = q.w[e];
            else if (scheme == 4) wgt = 0.75 * q.w[e] + 0.25 * p0;          // LUST<Type>::weights
            else {
                const double Nn = vf[nb];
                const double dx = Cx[nb] - Cx[c], dy = Cy[nb] - Cy[c], dz = Cz[nb] - Cz[c];
                const double gradf = Nn - P;
                const int up = flux > 0 ? c : nb;
                const double gradcf = dx * gx[up] + dy * gy[up] + dz * gz[up];
                double r;
                if (fabs(gradcf) >= 1000.0 * fabs(gradf)) {
                    // OpenFOAM's sign(): (s >= 0) ? 1 : -1, never 0 -- a uniform region gets r = 1999, limiter 1, linear weights
                    const double sa = gradcf >= 0 ? 1.0 : -1.0, sb = gradf >= 0 ? 1.0 : -1.0;
                    r = 2.0 * 1000.0 * sa * sb - 1.0;
                } else r = 2.0 * (gradcf / gradf) - 1.0;
                double lim = fmax(fmin(twoByk * r, 1.0), 0.0);
                if (scheme == 3) {
                    if ((flux > 0 && (P < lo || Nn > hi)) || (flux < 0 && (Nn < lo || P > hi))) lim = 0.0;
                }
                wgt = MODE == 0 ? lim * q.w[e] + (1.0 - lim) * p0 : lim;
            }
            out[e] = MODE == 2 ? fmin(out[e], wgt) : wgt;
        }
    }
}

// filteredLinear2V k l (cases/wallFireSpread2D/system/fvSchemes:41, `div(phi,U)`): one limiter per face for the three components of
// a vector field, from the face difference of the vector and twice the two cells' gradients projected on it
// (filteredLinear2VLimiter<NVDVTVDV>::limiter; oracle/fv.py: filtered_linear2V_weights)
struct FL2V { const double *U[3], *gx[3], *gy[3], *gz[3]; };
__global__ void k_filtered_linear2V_weights(MeshView q, double kk, double l1, const double *__restrict__ phi, FL2V a,
                                            const double *__restrict__ Cx, const double *__restrict__ Cy, const double *__restrict__ Cz,
                                            double *__restrict__ out)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci;
        const double P0 = a.U[0][c], P1 = a.U[1][c], P2 = a.U[2][c];
        FOR_OWN_FACES(q, c, e, nb) {
            const double g0 = a.U[0][nb] - P0, g1 = a.U[1][nb] - P1, g2 = a.U[2][nb] - P2;      // gradfV
            const double df = (g0 * g0 + g1 * g1) + g2 * g2;
            const double dx = Cx[nb] - Cx[c], dy = Cy[nb] - Cy[c], dz = Cz[nb] - Cz[c];
            double tc[2];
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int cc = s ? nb : c;
                const double d0 = (dx * a.gx[0][cc] + dy * a.gy[0][cc]) + dz * a.gz[0][cc];
                const double d1 = (dx * a.gx[1][cc] + dy * a.gy[1][cc]) + dz * a.gz[1][cc];
                const double d2 = (dx * a.gx[2][cc] + dy * a.gy[2][cc]) + dz * a.gz[2][cc];
                tc[s] = 2.0 * ((g0 * d0 + g1 * d1) + g2 * d2);
            }
            const double den = fmax(fabs(tc[0]), fabs(tc[1])) + 1.0e-15;
            double lim = (df > 0) ? l1 - kk * fmin(fmax(df - tc[0], 0.0), fmax(df - tc[1], 0.0)) / den
                                  : l1 - kk * fmin(fmax(tc[0] - df, 0.0), fmax(tc[1] - df, 0.0)) / den;
            lim = fmax(fmin(lim, 1.0), 0.0);
            const double p0 = phi[e] >= 0 ? 1.0 : 0.0;
            out[e] = lim * q.w[e] + (1.0 - lim) * p0;
        }
    }
}

// LUST<Type>::correction = 0.25 * linearUpwind<Type>::correction: (Cf - C_c) & grad(vf)_c, c = owner if the flux is > 0, else
// the neighbour (one scalar component; zero on non-coupled patches, which is what the boundary part of the caller holds)
__global__ void k_lust_correction(MeshView q, const double *__restrict__ phi, const double *__restrict__ gx, const double *__restrict__ gy,
                                  const double *__restrict__ gz, const double *__restrict__ Cx, const double *__restrict__ Cy,
                                  const double *__restrict__ Cz, const double *__restrict__ Cfx, const double *__restrict__ Cfy,
                                  const double *__restrict__ Cfz, double factor, double *__restrict__ out)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci;
        FOR_OWN_FACES(q, c, e, nb) {
            const int up = phi[e] > 0 ? c : nb;
            const double dx = Cfx[e] - Cx[up], dy = Cfy[e] - Cy[up], dz = Cfz[e] - Cz[up];
            out[e] = factor * ((dx * gx[up] + dy * gy[up]) + dz * gz[up]);       // factor 1: linearUpwind itself (1.0*x == x)
        }
    }
}

// fvMatrix<Type>::relax(alpha), non-coupled patches (coupled faces of a decomposed block are ordinary faces towards ghost
// cells here and enter sumMagOffDiag like internal faces).  One thread per row, sums in face order.
template <int W>
__global__ void k_relax(MeshView q, double alpha, int nc, const double *__restrict__ upper, const double *__restrict__ lower,
                        const double *__restrict__ ic0, const double *__restrict__ ic1, const double *__restrict__ ic2,
                        double *__restrict__ diag, const double *__restrict__ p0, const double *__restrict__ p1,
                        const double *__restrict__ p2, double *__restrict__ s0, double *__restrict__ s1, double *__restrict__ s2)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci;
        RowEnt<W> L, U; load_lower<W>(q.v, c, L); load_upper<W>(q.v, c, U);
        double sumOff = 0.0;
#pragma unroll
        for (int s = 0; s < W; s++) if (L.on[s]) sumOff += fabs(lower[L.f[s]]);
#pragma unroll
        for (int s = 0; s < W; s++) if (U.on[s]) sumOff += fabs(upper[U.f[s]]);
        const double D0 = diag[c];
        double D = D0;
        const int j = q.cellB[c];
        if (j >= 0) for (int t = q.bcStart[j]; t < q.bcStart[j + 1]; t++) {
            const int k = q.bcItem[t];
            D += (nc == 3) ? fmax(fmax(fabs(ic0[k]), fabs(ic1[k])), fabs(ic2[k])) : fabs(ic0[k]);
        }
        D = fmax(fabs(D), sumOff);
        D = D / alpha;
        if (j >= 0) for (int t = q.bcStart[j]; t < q.bcStart[j + 1]; t++) {
            const int k = q.bcItem[t];
            D -= (nc == 3) ? fmin(fmin(ic0[k], ic1[k]), ic2[k]) : ic0[k];
        }
        diag[c] = D;
        const double dd = D - D0;
        s0[c] += dd * p0[c];
        if (nc == 3) { s1[c] += dd * p1[c]; s2[c] += dd * p2[c]; }
    }
}

// ------------------------------------------------------------------ cell-sum kernels ---
// MODE 0: fvc::surfaceIntegrate (owner +, neighbour -, boundary +, then /V)   == fvc::div(ssf)
// MODE 1: fvc::surfaceSum       (owner +, neighbour +, boundary +)
template <int MODE, int W>
__global__ void k_face_sum(MeshView q, const double *__restrict__ ssf, const double *__restrict__ ssb, double *__restrict__ out)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci;
        RowEnt<W> L, U; load_lower<W>(q.v, c, L); load_upper<W>(q.v, c, U);
        double a[W], b[W];
#pragma unroll
        for (int s = 0; s < W; s++) { a[s] = ssf[L.f[s]]; b[s] = ssf[U.f[s]]; }
        // the reference's face loop touches a cell's faces in face order: faces where it is the
        // neighbour come first, then the faces it owns
        double acc = 0.0;
#pragma unroll
        for (int s = 0; s < W; s++) if (L.on[s]) acc = (MODE == 0) ? acc - a[s] : acc + a[s];
#pragma unroll
        for (int s = 0; s < W; s++) if (U.on[s]) acc += b[s];
        const int j = q.cellB[c];
        if (j >= 0 && ssb) for (int t = q.bcStart[j]; t < q.bcStart[j + 1]; t++) acc += ssb[q.bcItem[t]];
        out[c] = (MODE == 0) ? acc / q.V[c] : acc;
    }
}

// fvc::grad, Gauss linear: (1/V) sum_f Sf * (w P + (1-w) N); face values formed on the fly exactly
// as the owner row would form them, so no face field is materialised
template <int W>
__global__ void k_grad(MeshView q, const double *__restrict__ vf, const double *__restrict__ vb, double *__restrict__ gx,
                       double *__restrict__ gy, double *__restrict__ gz)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci;
        RowEnt<W> L, U; load_lower<W>(q.v, c, L); load_upper<W>(q.v, c, U);
        const double P = vf[c];
        double vl[W], vu[W], wl[W], wu[W], sx[2 * W], sy[2 * W], sz[2 * W];
#pragma unroll
        for (int s = 0; s < W; s++) {
            vl[s] = vf[L.nb[s]]; vu[s] = vf[U.nb[s]]; wl[s] = q.w[L.f[s]]; wu[s] = q.w[U.f[s]];
            sx[s] = q.Sfx[L.f[s]]; sy[s] = q.Sfy[L.f[s]]; sz[s] = q.Sfz[L.f[s]];
            sx[W + s] = q.Sfx[U.f[s]]; sy[W + s] = q.Sfy[U.f[s]]; sz[W + s] = q.Sfz[U.f[s]];
        }
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
        const int j = q.cellB[c];
        if (j >= 0) for (int t = q.bcStart[j]; t < q.bcStart[j + 1]; t++) {
            const int k = q.bcItem[t]; const double b = vb[k];
            ax += q.bSfx[k] * b; ay += q.bSfy[k] * b; az += q.bSfz[k] * b;
        }
        const double V = q.V[c];
        gx[c] = ax / V; gy[c] = ay / V; gz[c] = az / V;
    }
}

// fvc::reconstruct(ssf) = inv(surfaceSum(Sf (x) Sf/magSf)) & surfaceSum((Sf/magSf) ssf)
template <int W>
__global__ void k_reconstruct(MeshView q, long invStride, const double *__restrict__ invT, const double *__restrict__ bMagSf,
                              const double *__restrict__ ssf, const double *__restrict__ ssb, double *__restrict__ ox,
                              double *__restrict__ oy, double *__restrict__ oz)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci; const long N = invStride;      // invT is [6][nCells] (owned + ghost)
        RowEnt<W> L, U; load_lower<W>(q.v, c, L); load_upper<W>(q.v, c, U);
        double vx = 0, vy = 0, vz = 0;
#pragma unroll
        for (int s = 0; s < W; s++) if (L.on[s]) {
            const int e = L.f[s]; const double t = ssf[e], mg = q.magSf[e];
            vx += q.Sfx[e] / mg * t; vy += q.Sfy[e] / mg * t; vz += q.Sfz[e] / mg * t;
        }
#pragma unroll
        for (int s = 0; s < W; s++) if (U.on[s]) {
            const int e = U.f[s]; const double t = ssf[e], mg = q.magSf[e];
            vx += q.Sfx[e] / mg * t; vy += q.Sfy[e] / mg * t; vz += q.Sfz[e] / mg * t;
        }
        const int j = q.cellB[c];
        if (j >= 0 && ssb) for (int t = q.bcStart[j]; t < q.bcStart[j + 1]; t++) {
            const int k = q.bcItem[t]; const double b = ssb[k], mg = bMagSf[k];
            vx += q.bSfx[k] / mg * b; vy += q.bSfy[k] / mg * b; vz += q.bSfz[k] / mg * b;
        }
        const double xx = invT[c], xy = invT[N + c], xz = invT[2 * N + c], yy = invT[3 * N + c], yz = invT[4 * N + c], zz = invT[5 * N + c];
        ox[c] = xx * vx + xy * vy + xz * vz; oy[c] = xy * vx + yy * vy + yz * vz; oz[c] = xz * vx + yz * vy + zz * vz;
    }
}

// ------------------------------------------------------------------ fvm assembly ---
// One kernel assembles a transport matrix  [fvm::ddt(rho, .)] + [fvm::div(phi, .)] - [fvm::laplacian(gamma, .)]
// (any bracket may be absent: pass NULL).  Face coefficients are written by the owner row; the
// diagonal applies negSumDiag per term in the reference's face order and then combines the terms
// in the order ddt + div - laplacian, exactly as the reference's tmp<fvMatrix> algebra does.
template <int W>
__global__ void k_fvm_transport(MeshView q, double rDeltaT, const double *__restrict__ rho, const double *__restrict__ phi,
                                const double *__restrict__ wf, const double *__restrict__ gamma, double lapSign,
                                double *__restrict__ diag, double *__restrict__ upper, double *__restrict__ lower)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci;
        RowEnt<W> L, U; load_lower<W>(q.v, c, L); load_upper<W>(q.v, c, U);
        double dDiv = 0.0, dLap = 0.0;
        // faces where c is the neighbour: diag -= upper[f]
#pragma unroll
        for (int s = 0; s < W; s++) if (L.on[s]) {
            const int e = L.f[s];
            if (phi) { const double f = phi[e]; const double lo = -wf[e] * f; dDiv -= (lo + f); }
            if (gamma) dLap -= gamma[e] * q.magSf[e] * q.delta[e];
        }
        // faces owned by c: write coefficients, diag -= lower[f]
#pragma unroll
        for (int s = 0; s < W; s++) if (U.on[s]) {
            const int e = U.f[s];
            double lo = 0.0, up = 0.0;
            if (phi) { const double f = phi[e]; lo = -wf[e] * f; up = lo + f; dDiv -= lo; }
            if (gamma) {
                const double g = gamma[e] * q.magSf[e] * q.delta[e];
                dLap -= g;
                if (phi) { lo = lapSign < 0 ? lo - g : lo + g; up = lapSign < 0 ? up - g : up + g; }
                else { lo = lapSign < 0 ? -g : g; up = lo; }
            }
            if (upper) upper[e] = up;
            if (lower) lower[e] = lo;
        }
        double d = rho ? rDeltaT * rho[c] * q.V[c] : 0.0;
        if (phi) d = rho ? d + dDiv : dDiv;
        if (gamma) d = (rho || phi) ? (lapSign < 0 ? d - dLap : d + dLap) : (lapSign < 0 ? -dLap : dLap);
        diag[c] = d;
    }
}

// boundary coefficients of [div] - [laplacian] for a field whose patch condition is `mixed` (f, ref, refGrad):
//   internalCoeffs = phib*(1-f) -/+ gammab*magSf*(-f*delta)
//   boundaryCoeffs = -phib*(f*ref + (1-f)*refGrad/delta) +/- gammab*magSf*(f*delta*ref + (1-f)*refGrad)
__global__ void k_boundary_coeffs(int B, const double *__restrict__ phib, const double *__restrict__ gammab, double lapSign,
                                  const double *__restrict__ magSf, const double *__restrict__ delta, const double *__restrict__ f,
                                  const double *__restrict__ ref, const double *__restrict__ refGrad, double *__restrict__ ic,
                                  double *__restrict__ bc)
{
    GRID_STRIDE(k, B) {
        const double fk = f[k], rk = ref[k], gk = refGrad[k], dk = delta[k];
        double i = 0.0, b = 0.0;
        if (phib) { i = phib[k] * (1.0 - fk); b = -phib[k] * (fk * rk + (1.0 - fk) * gk / dk); }
        if (gammab) {
            const double pG = gammab[k] * magSf[k];
            const double li = pG * (-fk * dk), lb = -pG * (fk * dk * rk + (1.0 - fk) * gk);
            if (phib) { i = lapSign < 0 ? i - li : i + li; b = lapSign < 0 ? b - lb : b + lb; }
            else { i = lapSign < 0 ? -li : li; b = lapSign < 0 ? -lb : lb; }
        }
        ic[k] = i; bc[k] = b;
    }
}

// value of a mixed patch field: f*ref + (1-f)*(cell + refGrad/delta)
__global__ void k_bc_values(int B, const int *__restrict__ fc, const double *__restrict__ delta, const double *__restrict__ f,
                            const double *__restrict__ ref, const double *__restrict__ refGrad, const double *__restrict__ vf,
                            double *__restrict__ out)
{ GRID_STRIDE(k, B) out[k] = f[k] * ref[k] + (1.0 - f[k]) * (vf[fc[k]] + refGrad[k] / delta[k]); }

// addBoundaryDiag + addBoundarySource (+ optional explicit volume source V*su):
//   diagOut = diag + sum ic ; srcOut = src + sum bc (+ V*su)
__global__ void k_add_boundary(MeshView q, const double *__restrict__ ic, const double *__restrict__ bc,
                               const double *__restrict__ diag, const double *__restrict__ src, const double *__restrict__ su,
                               double *__restrict__ diagOut, double *__restrict__ srcOut)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci;
        double d = diag ? diag[c] : 0.0, s = src ? src[c] : 0.0;
        const int j = q.cellB[c];
        if (j >= 0) for (int t = q.bcStart[j]; t < q.bcStart[j + 1]; t++) { const int k = q.bcItem[t]; if (ic) d += ic[k]; if (bc) s += bc[k]; }
        if (su) s += q.V[c] * su[c];
        if (diagOut) diagOut[c] = d;
        if (srcOut) srcOut[c] = s;
    }
}

// fvMatrix::A(): (diag + cmptAv(internalCoeffs))/V for nc components
__global__ void k_matrix_A(MeshView q, int nc, const double *__restrict__ diag, const double *__restrict__ ic0,
                           const double *__restrict__ ic1, const double *__restrict__ ic2, double *__restrict__ out)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci;
        double d = diag[c];
        const int j = q.cellB[c];
        if (j >= 0) for (int t = q.bcStart[j]; t < q.bcStart[j + 1]; t++) {
            const int k = q.bcItem[t];
            d += (nc == 3) ? (ic0[k] + ic1[k] + ic2[k]) / 3.0 : ic0[k];
        }
        out[c] = d / q.V[c];
    }
}

// fvMatrix::H() for one component: ((avgBD - BD_cmpt)*psi - sum_offdiag a*psi_nb + source + boundarySource)/V
template <int W>
__global__ void k_matrix_H(MeshView q, int nc, const double *__restrict__ upper, const double *__restrict__ lower,
                           const double *__restrict__ src, const double *__restrict__ icC, const double *__restrict__ ic0,
                           const double *__restrict__ ic1, const double *__restrict__ ic2, const double *__restrict__ bcC,
                           const double *__restrict__ psi, double *__restrict__ out)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci;
        RowEnt<W> L, U; load_lower<W>(q.v, c, L); load_upper<W>(q.v, c, U);
        double al[W], au[W], xl[W], xu[W];
#pragma unroll
        for (int s = 0; s < W; s++) { al[s] = lower[L.f[s]]; au[s] = upper[U.f[s]]; xl[s] = psi[L.nb[s]]; xu[s] = psi[U.nb[s]]; }
        double bd = 0.0, bda = 0.0, bs = 0.0;
        const int j = q.cellB[c];
        if (j >= 0) for (int t = q.bcStart[j]; t < q.bcStart[j + 1]; t++) {
            const int k = q.bcItem[t];
            bd += icC[k]; bda += (nc == 3) ? (ic0[k] + ic1[k] + ic2[k]) / 3.0 : ic0[k]; bs += bcC[k];
        }
        double hl = 0.0;
#pragma unroll
        for (int s = 0; s < W; s++) if (L.on[s]) hl -= al[s] * xl[s];
#pragma unroll
        for (int s = 0; s < W; s++) if (U.on[s]) hl -= au[s] * xu[s];
        double h = (bda - bd) * psi[c];
        h += hl + src[c];
        h += bs;
        out[c] = h / q.V[c];
    }
}

// fvMatrix::flux(): internal upper*psi_u - lower*psi_l ; boundary internalCoeffs*psi_c - boundaryCoeffs
__global__ void k_matrix_flux(MeshView q, const double *__restrict__ upper, const double *__restrict__ lower,
                              const double *__restrict__ psi, double *__restrict__ out)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci; const double P = psi[c];
        FOR_OWN_FACES(q, c, e, nb) out[e] = upper[e] * psi[nb] - lower[e] * P;
    }
}
__global__ void k_matrix_flux_b(int B, const int *__restrict__ fc, const double *__restrict__ ic, const double *__restrict__ bc,
                                const double *__restrict__ psi, double *__restrict__ out)
{ GRID_STRIDE(k, B) out[k] = ic[k] * psi[fc[k]] - bc[k]; }

// ------------------------------------------------------------------ C ABI ---
#define CHECK_M(m) if (!(m)) { ffm_set_error("null mesh"); return FFM_ERR_ARG; }
#define DONE() FFM_HIP(hipGetLastError()); return FFM_OK

extern "C" int ffm_fvc_interpolate(ffm_mesh *m, const double *w_f, const double *vf, double *out_f)
{ CHECK_M(m); LAUNCH_CELLS(k_interpolate, mview(m), w_f, vf, out_f); DONE(); }
extern "C" int ffm_fvc_snGrad(ffm_mesh *m, const double *vf, double *out_f)
{ CHECK_M(m); LAUNCH_CELLS(k_snGrad, mview(m), vf, out_f); DONE(); }
extern "C" int ffm_fvc_snGrad_b(ffm_mesh *m, const double *vf, const double *vb, double *out_b)
{ CHECK_M(m); if (m->B) LAUNCH(k_snGrad_b, m->B, m->B, m->bCells, m->bDelta, vf, vb, out_b); DONE(); }
extern "C" int ffm_fvc_flux(ffm_mesh *m, const double *vx, const double *vy, const double *vz, double *out_f)
{ CHECK_M(m); LAUNCH_CELLS(k_flux, mview(m), vx, vy, vz, out_f); DONE(); }
extern "C" int ffm_fvc_surface_integrate(ffm_mesh *m, const double *ssf, const double *ssb, double *out)
{ CHECK_M(m); FFM_DISPATCH_W(m->A->maxW, LAUNCH_CELLS((k_face_sum<0, W>), mview(m), ssf, ssb, out)); DONE(); }
extern "C" int ffm_fvc_surface_sum(ffm_mesh *m, const double *ssf, const double *ssb, double *out)
{ CHECK_M(m); FFM_DISPATCH_W(m->A->maxW, LAUNCH_CELLS((k_face_sum<1, W>), mview(m), ssf, ssb, out)); DONE(); }
extern "C" int ffm_fvc_grad(ffm_mesh *m, const double *vf, const double *vb, double *gx, double *gy, double *gz)
{ CHECK_M(m); FFM_DISPATCH_W(m->A->maxW, LAUNCH_CELLS(k_grad<W>, mview(m), vf, vb, gx, gy, gz)); DONE(); }
extern "C" int ffm_fvc_reconstruct(ffm_mesh *m, const double *ssf, const double *ssb, double *ox, double *oy, double *oz)
{ CHECK_M(m); FFM_DISPATCH_W(m->A->maxW, LAUNCH_CELLS(k_reconstruct<W>, mview(m), (long)m->N, m->invT, m->bMagSf, ssf, ssb, ox, oy, oz)); DONE(); }
extern "C" int ffm_fv_limited_weights(ffm_mesh *m, int scheme, double k, double lo, double hi, const double *phi_f,
                                      const double *vf, const double *gx, const double *gy, const double *gz, double *out_w)
{
    CHECK_M(m);
    if (scheme < 0 || scheme > 5 || !phi_f || !out_w || ((scheme == 2 || scheme == 3) && (!vf || !gx || !gy || !gz))) return FFM_ERR_ARG;
    LAUNCH_CELLS(k_limited_weights<0>, mview(m), scheme, 2.0 / std::max(k, 1e-15), lo, hi, phi_f, vf, gx, gy, gz, m->C[0], m->C[1], m->C[2], out_w);
    DONE();
}
// The limiter of limitedLinear k (scheme 2) / limitedLinear01 k (scheme 3) for one field; accumulateMin != 0: lim_f = min(lim_f, limiter).
// With ffm_fv_weights_from_limiter this is multivariateSelectionScheme (solver/YEEqn.H:1-10, cases/steckler/system/fvSchemes:36-47):
// one limiter for all fields of the table -- the face-wise minimum of the member schemes' limiters -- and the weights made of it.
extern "C" int ffm_fv_limited_limiter(ffm_mesh *m, int scheme, double k, double lo, double hi, const double *phi_f, const double *vf,
                                      const double *gx, const double *gy, const double *gz, double *lim_f, int accumulateMin)
{
    CHECK_M(m);
    if ((scheme != 2 && scheme != 3) || !phi_f || !lim_f || !vf || !gx || !gy || !gz) return FFM_ERR_ARG;
    if (accumulateMin) LAUNCH_CELLS(k_limited_weights<2>, mview(m), scheme, 2.0 / std::max(k, 1e-15), lo, hi, phi_f, vf, gx, gy, gz, m->C[0], m->C[1], m->C[2], lim_f);
    else LAUNCH_CELLS(k_limited_weights<1>, mview(m), scheme, 2.0 / std::max(k, 1e-15), lo, hi, phi_f, vf, gx, gy, gz, m->C[0], m->C[1], m->C[2], lim_f);
    DONE();
}
__global__ void k_weights_from_limiter(MeshView q, const double *__restrict__ phi, const double *__restrict__ lim, double *__restrict__ out)
{
    CELL_SCHED(ci, q) {
        const int c = (int)ci;
        FOR_OWN_FACES(q, c, e, nb) { (void)nb; const double l = lim[e], p0 = phi[e] >= 0 ? 1.0 : 0.0; out[e] = l * q.w[e] + (1.0 - l) * p0; }
    }
}
// limitedSurfaceInterpolationScheme::weights from a limiter field: limiter*linear + (1 - limiter)*upwind
extern "C" int ffm_fv_weights_from_limiter(ffm_mesh *m, const double *phi_f, const double *lim_f, double *out_w)
{
    CHECK_M(m);
    if (!phi_f || !lim_f || !out_w) return FFM_ERR_ARG;
    LAUNCH_CELLS(k_weights_from_limiter, mview(m), phi_f, lim_f, out_w);
    DONE();
}
extern "C" int ffm_fv_filtered_linear2V_weights(ffm_mesh *m, double k, double l, const double *phi_f, const double *const *U,
                                                const double *const *gx, const double *const *gy, const double *const *gz, double *out_w)
{
    CHECK_M(m);
    if (!phi_f || !U || !gx || !gy || !gz || !out_w || k < 0 || k > 1 || l < 0 || l > 1) return FFM_ERR_ARG;
    FL2V a;
    for (int d = 0; d < 3; d++) {
        if (!U[d] || !gx[d] || !gy[d] || !gz[d]) return FFM_ERR_ARG;
        a.U[d] = U[d]; a.gx[d] = gx[d]; a.gy[d] = gy[d]; a.gz[d] = gz[d];
    }
    LAUNCH_CELLS(k_filtered_linear2V_weights, mview(m), k, l + 1.0, phi_f, a, m->C[0], m->C[1], m->C[2], out_w);
    DONE();
}
extern "C" int ffm_mesh_set_face_centres(ffm_mesh *m, const double *Cf)
{
    CHECK_M(m);
    if (!Cf) return FFM_ERR_ARG;
    std::vector<double> v(std::max(m->nNat, 1));          // one staging array, filled and scattered by the host threads
    for (int d = 0; d < 3; d++) {
        const int *c2n = m->A->h_callerToNative.data(); double *vp = v.data(); const double *src = Cf + (size_t)d * m->F;
        ffm_parallel_for(m->nNat, [=](long lo, long hi) { for (long q = lo; q < hi; q++) vp[q] = 0.0; });
        ffm_parallel_for(m->F, [=](long lo, long hi) { for (long f = lo; f < hi; f++) vp[c2n[f]] = src[f]; });
        hipFree(m->Cf[d]); m->Cf[d] = nullptr;
        FFM_TRY(up(m->ctx, &m->Cf[d], v));
    }
    return FFM_OK;
}
// mesh.nonOrthCorrectionVectors() of the internal faces, host array [3][F] in LDU face order (zero on orthogonal meshes, where
// this call is not needed; non-coupled boundary faces carry no correction upstream)
extern "C" int ffm_mesh_set_nonorth_correction(ffm_mesh *m, const double *corrVec)
{
    CHECK_M(m);
    if (!corrVec) return FFM_ERR_ARG;
    std::vector<double> v(std::max(m->nNat, 1));          // one staging array, filled and scattered by the host threads
    for (int d = 0; d < 3; d++) {
        const int *c2n = m->A->h_callerToNative.data(); double *vp = v.data(); const double *src = corrVec + (size_t)d * m->F;
        ffm_parallel_for(m->nNat, [=](long lo, long hi) { for (long q = lo; q < hi; q++) vp[q] = 0.0; });
        ffm_parallel_for(m->F, [=](long lo, long hi) { for (long f = lo; f < hi; f++) vp[c2n[f]] = src[f]; });
        hipFree(m->corr[d]); m->corr[d] = nullptr;
        FFM_TRY(up(m->ctx, &m->corr[d], v));
    }
    return FFM_OK;
}
// correctedSnGrad::correction(vf): out_f = nonOrthCorrectionVectors & interpolate(grad(vf)); `Gauss linear corrected` adds
//   fvc::snGrad:      snGrad = uncorrected + out_f
//   fvm::laplacian:   source -= V * ffm_fvc_surface_integrate(gamma_f*magSf*out_f, 0)       (gaussLaplacianScheme::fvmLaplacian)
extern "C" int ffm_fvc_snGrad_correction(ffm_mesh *m, const double *gx, const double *gy, const double *gz, double *out_f)
{
    CHECK_M(m);
    if (!gx || !gy || !gz || !out_f) return FFM_ERR_ARG;
    if (!m->corr[0]) { ffm_set_error("ffm_fvc_snGrad_correction: correction vectors not set (ffm_mesh_set_nonorth_correction)"); return FFM_ERR_ARG; }
    LAUNCH_CELLS(k_snGrad_correction, mview(m), m->corr[0], m->corr[1], m->corr[2], gx, gy, gz, out_f);
    DONE();
}
extern "C" int ffm_fv_lust_correction(ffm_mesh *m, const double *phi_f, const double *gx, const double *gy, const double *gz, double *out_f)
{
    CHECK_M(m);
    if (!phi_f || !gx || !gy || !gz || !out_f) return FFM_ERR_ARG;
    if (!m->Cf[0]) { ffm_set_error("ffm_fv_lust_correction: face centres not set (ffm_mesh_set_face_centres)"); return FFM_ERR_ARG; }
    LAUNCH_CELLS(k_lust_correction, mview(m), phi_f, gx, gy, gz, m->C[0], m->C[1], m->C[2], m->Cf[0], m->Cf[1], m->Cf[2], 0.25, out_f);
    DONE();
}
extern "C" int ffm_fv_linear_upwind_correction(ffm_mesh *m, const double *phi_f, const double *gx, const double *gy, const double *gz, double *out_f)
{
    CHECK_M(m);
    if (!phi_f || !gx || !gy || !gz || !out_f) return FFM_ERR_ARG;
    if (!m->Cf[0]) { ffm_set_error("ffm_fv_linear_upwind_correction: face centres not set (ffm_mesh_set_face_centres)"); return FFM_ERR_ARG; }
    LAUNCH_CELLS(k_lust_correction, mview(m), phi_f, gx, gy, gz, m->C[0], m->C[1], m->C[2], m->Cf[0], m->Cf[1], m->Cf[2], 1.0, out_f);
    DONE();
}
extern "C" int ffm_fvm_relax(ffm_mesh *m, double alpha, int nc, const double *upper, const double *lower, const double *ic0,
                             const double *ic1, const double *ic2, double *diag, const double *psi0, const double *psi1,
                             const double *psi2, double *src0, double *src1, double *src2)
{
    CHECK_M(m);
    if ((nc != 1 && nc != 3) || !upper || !lower || !diag || !psi0 || !src0 || (m->B && !ic0)) return FFM_ERR_ARG;
    if (nc == 3 && (!psi1 || !psi2 || !src1 || !src2 || (m->B && (!ic1 || !ic2)))) return FFM_ERR_ARG;
    if (alpha <= 0) return FFM_OK;                              // fvMatrix::relax: no-op
    FFM_DISPATCH_W(m->A->maxW, LAUNCH_CELLS(k_relax<W>, mview(m), alpha, nc, upper, lower, ic0, ic1, ic2, diag, psi0, psi1, psi2, src0, src1, src2));
    DONE();
}
extern "C" int ffm_fvm_transport(ffm_mesh *m, double rDeltaT, const double *rho, const double *phi_f, const double *w_f,
                                 const double *gamma_f, int laplacianSign, double *diag, double *upper, double *lower)
{
    CHECK_M(m);
    // upper == NULL: no face coefficients wanted (a pure ddt term has none); lower == NULL: a symmetric result (no convection), lower = upper
    if (!diag || ((phi_f || gamma_f) && !upper) || (phi_f && (!w_f || !lower))) return FFM_ERR_ARG;
    FFM_DISPATCH_W(m->A->maxW, LAUNCH_CELLS(k_fvm_transport<W>, mview(m), rDeltaT, rho, phi_f, w_f, gamma_f, (double)laplacianSign, diag, upper, lower));
    DONE();
}
extern "C" int ffm_fvm_boundary_coeffs(ffm_mesh *m, const double *phib, const double *gammab, int laplacianSign, const double *f,
                                       const double *ref, const double *refGrad, double *ic, double *bc)
{ CHECK_M(m); if (m->B) LAUNCH(k_boundary_coeffs, m->B, m->B, phib, gammab, (double)laplacianSign, m->bMagSf, m->bDelta, f, ref, refGrad, ic, bc); DONE(); }
extern "C" int ffm_bc_values(ffm_mesh *m, const double *f, const double *ref, const double *refGrad, const double *vf, double *out_b)
{ CHECK_M(m); if (m->B) LAUNCH(k_bc_values, m->B, m->B, m->bCells, m->bDelta, f, ref, refGrad, vf, out_b); DONE(); }
extern "C" int ffm_fvm_add_boundary(ffm_mesh *m, const double *ic, const double *bc, const double *diag, const double *src,
                                    const double *su, double *diagOut, double *srcOut)
{ CHECK_M(m); LAUNCH_CELLS(k_add_boundary, mview(m), ic, bc, diag, src, su, diagOut, srcOut); DONE(); }
extern "C" int ffm_fvm_A(ffm_mesh *m, int nc, const double *diag, const double *ic0, const double *ic1, const double *ic2, double *out)
{ CHECK_M(m); if (nc != 1 && nc != 3) return FFM_ERR_ARG; LAUNCH_CELLS(k_matrix_A, mview(m), nc, diag, ic0, ic1, ic2, out); DONE(); }
extern "C" int ffm_fvm_H(ffm_mesh *m, int nc, int cmpt, const double *upper, const double *lower, const double *src,
                         const double *ic0, const double *ic1, const double *ic2, const double *bcC, const double *psi, double *out)
{
    CHECK_M(m); if ((nc != 1 && nc != 3) || cmpt < 0 || cmpt >= nc) return FFM_ERR_ARG;
    const double *icC = cmpt == 0 ? ic0 : cmpt == 1 ? ic1 : ic2;
    FFM_DISPATCH_W(m->A->maxW, LAUNCH_CELLS(k_matrix_H<W>, mview(m), nc, upper, lower, src, icC, ic0, ic1, ic2, bcC, psi, out));
    DONE();
}
extern "C" int ffm_fvm_flux(ffm_mesh *m, const double *upper, const double *lower, const double *ic, const double *bc,
                            const double *psi, double *out_f, double *out_b)
{
    CHECK_M(m);
    if (out_f) LAUNCH_CELLS(k_matrix_flux, mview(m), upper, lower, psi, out_f);      // (out_f null: the boundary part only)
    if (m->B && out_b) LAUNCH(k_matrix_flux_b, m->B, m->B, m->bCells, ic, bc, psi, out_b);
    DONE();
}

extern "C" const double *ffm_mesh_geometry_d(const ffm_mesh *m, int which)
{
    if (!m) return nullptr;
    switch (which) {
    case 0: return m->V; case 1: return m->magSf; case 2: return m->delta; case 3: return m->w; case 4: return m->bMagSf; case 5: return m->bDelta;
    case 6: return m->bSf[0]; case 7: return m->bSf[1]; case 8: return m->bSf[2];
    case 9: return m->Sf[0]; case 10: return m->Sf[1]; case 11: return m->Sf[2]; case 12: return m->C[0]; case 13: return m->C[1]; case 14: return m->C[2];
    }
    return nullptr;
}

// internal accessors for the case driver
const int *ffm_mesh_bcells(const ffm_mesh *m) { return m->bCells; }
const double *ffm_mesh_geom(const ffm_mesh *m, int which)
{
    switch (which) {
    case 0: return m->V; case 1: return m->magSf; case 2: return m->delta; case 3: return m->w;
    case 4: return m->bMagSf; case 5: return m->bDelta; case 6: return m->bSf[0]; case 7: return m->bSf[1]; case 8: return m->bSf[2];
    case 9: return m->Sf[0]; case 10: return m->Sf[1]; case 11: return m->Sf[2]; case 12: return m->C[0]; case 13: return m->C[1]; case 14: return m->C[2];
    }
    return nullptr;
}
