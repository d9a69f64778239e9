// ffm_rad.hip -- SURVEY 8(f) N1: the wall condition of the fvDOM rays, greyDiffusiveRadiationMixedFvPatchScalarField::updateCoeffs
// (reference packages/thermophysicalModels/radiation/derivedFvPatchFields/greyDiffusiveRadiation/
// greyDiffusiveRadiationMixedFvPatchScalarField.C:150-230), for ALL boundary faces of the mesh in one pass per ray solve.
// The ray equations themselves are assembled and solved by the transport kernels (ffm_fv.hip) and the LDU solvers; what is
// specific to the radiation model is this coupling of the rays through the walls: a wall of emissivity e sends back
// (1 - e) of the radiation the OTHER rays deposit on it (their qin), which is what fvDOM::calculate iterates on
// (fvDOM/fvDOM.C:547-584; cases/wallFireSpread2D/constant/radiationProperties:39-46: maxIter 5, convergence 1e-3).
#include "ffm_mesh.hpp"

// one thread per boundary face k; ray arrays are [nRay][B].  Order of the Ir sum: rays 0 .. nRay-1, the ray's own qin counted as
// the zero it was reset to (radiativeIntensityRay::correct: qin_.boundaryFieldRef() = 0 before the matrix is built).
__global__ void k_grey_diffusive(int B, int nRay, int ray, double dx, double dy, double dz, double ax, double ay, double az, double sigma,
                                 const double *__restrict__ bSx, const double *__restrict__ bSy, const double *__restrict__ bSz,
                                 const double *__restrict__ bMag, const double *__restrict__ Iw, const double *__restrict__ emis,
                                 const double *__restrict__ Tb, double *__restrict__ qinAll, double *__restrict__ qemAll,
                                 double *__restrict__ qrAll, double *__restrict__ f, double *__restrict__ ref)
{
    GRID_STRIDE(k, B) {
        const double nx = bSx[k] / bMag[k], ny = bSy[k] / bMag[k], nz = bSz[k] / bMag[k];
        const double nAve = (nx * ax + ny * ay) + nz * az;
        const double iw = Iw[k];
        qrAll[(size_t)ray * B + k] = 0.0 + iw * nAve;
        double Ir = ray == 0 ? 0.0 : qinAll[k];
        for (int j = 1; j < nRay; j++) Ir = Ir + (j == ray ? 0.0 : qinAll[(size_t)j * B + k]);
        const bool out = -((nx * dx + ny * dy) + nz * dz) > 0.0;
        const double e = emis ? emis[k] : 1.0, t = Tb[k];
        const double val = (Ir * (1.0 - e) + e * sigma * ((t * t) * (t * t))) / M_PI;
        f[k] = out ? 1.0 : 0.0;
        ref[k] = out ? val : 0.0;
        qemAll[(size_t)ray * B + k] = out ? val * nAve : 0.0;
        qinAll[(size_t)ray * B + k] = out ? 0.0 : iw * nAve;
    }
}

extern "C" int ffm_fvdom_wall_coeffs_d(ffm_mesh *m, int nRay, int ray, const double *d3, const double *dAve3, double sigma, const double *Iw_b,
                                       const double *emissivity_b, const double *T_b, double *qin_all, double *qem_all, double *qr_all,
                                       double *f_b, double *ref_b)
{
    if (!m || !d3 || !dAve3 || !Iw_b || !T_b || !qin_all || !qem_all || !qr_all || !f_b || !ref_b || nRay < 1 || ray < 0 || ray >= nRay) return FFM_ERR_ARG;
    if (m->B == 0) return FFM_OK;
    LAUNCH(k_grey_diffusive, m->B, m->B, nRay, ray, d3[0], d3[1], d3[2], dAve3[0], dAve3[1], dAve3[2], sigma, m->bSf[0], m->bSf[1], m->bSf[2], m->bMagSf,
           Iw_b, emissivity_b, T_b, qin_all, qem_all, qr_all, f_b, ref_b);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}

// fvDOM::updateG, boundary part: out[k] = sum over the rays (in ray order) of all[ray][k]
__global__ void k_sum_rays(int B, int nRay, const double *__restrict__ all, double *__restrict__ out)
{
    GRID_STRIDE(k, B) {
        double s = 0.0;
        for (int j = 0; j < nRay; j++) s = s + all[(size_t)j * B + k];
        out[k] = s;
    }
}
extern "C" int ffm_fvdom_sum_rays_d(ffm_mesh *m, int nRay, const double *all, double *out_b)
{
    if (!m || !all || !out_b || nRay < 1) return FFM_ERR_ARG;
    if (m->B == 0) return FFM_OK;
    LAUNCH(k_sum_rays, m->B, m->B, nRay, all, out_b);
    FFM_HIP(hipGetLastError());
    return FFM_OK;
}
