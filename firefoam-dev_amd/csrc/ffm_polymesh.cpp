// ffm_polymesh.cpp -- SURVEY 8(f) N4, first part: reader of OpenFOAM's on-disk mesh, constant/polyMesh/{points,faces,owner,
// neighbour,boundary} (ascii), and the finite-volume geometry derived from it, so that a case directory produced by blockMesh /
// topoSet / createBaffles (reference cases/steckler/mesh.sh:8-21) can be handed to ffm_ldu_create / ffm_mesh_create.
// Host-only C++; nothing here touches the device.
//
// Geometry follows the published algorithms of OpenFOAM-dev (not vendored in the reference):
//   face centres / area vectors   primitiveMeshFaceCentresAndAreas.C   (triangle fan around the average point)
//   cell centres / volumes        primitiveMeshCellCentresAndVols.C    (pyramids on an estimated centre)
//   weights                       surfaceInterpolation::makeWeights
//   nonOrthDeltaCoeffs            surfaceInterpolation::makeNonOrthDeltaCoeffs      1/max(nf & d, 0.05|d|)
//   nonOrthCorrectionVectors      surfaceInterpolation::makeNonOrthCorrectionVectors   nf - d*nonOrthDeltaCoeffs
//   patch deltaCoeffs             fvPatch::deltaCoeffs   1/(nf & (Cf - C))
// File format: FoamFile header, then `N ( ... )`; faces as `n(v0 v1 ...)`; boundary as `N ( name { type ..; nFaces ..;
// startFace ..; } ... )`.  Binary files are refused.
#include "../../include/ffm.h"
#include <array>
#include <cctype>
#include <cerrno>
#include <algorithm>
#include <exception>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

void ffm_set_error(const char *fmt, ...);

namespace {
typedef std::array<double, 3> vec;
inline vec operator-(const vec &a, const vec &b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
inline vec operator+(const vec &a, const vec &b) { return {a[0] + b[0], a[1] + b[1], a[2] + b[2]}; }
inline vec operator*(double s, const vec &a) { return {s * a[0], s * a[1], s * a[2]}; }
inline double dot(const vec &a, const vec &b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline vec cross(const vec &a, const vec &b) { return {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}; }
inline double mag(const vec &a) { return std::sqrt(dot(a, a)); }

// tokens of an OpenFOAM ascii file: words / numbers, and the punctuation ( ) { } ; as single-character tokens
struct Tokens {
    std::vector<std::string> t; size_t i = 0;
    bool load(const std::string &path)
    {
        std::ifstream f(path);
        if (!f) { ffm_set_error("cannot open %s", path.c_str()); return false; }
        std::stringstream ss; ss << f.rdbuf();
        const std::string s = ss.str();
        for (size_t p = 0; p < s.size();) {
            const char c = s[p];
            if (std::isspace((unsigned char)c)) { p++; continue; }
            if (c == '/' && p + 1 < s.size() && s[p + 1] == '/') { while (p < s.size() && s[p] != '\n') p++; continue; }
            if (c == '/' && p + 1 < s.size() && s[p + 1] == '*') { p = s.find("*/", p + 2); p = (p == std::string::npos) ? s.size() : p + 2; continue; }
            if (c == '(' || c == ')' || c == '{' || c == '}' || c == ';') { t.push_back(std::string(1, c)); p++; continue; }
            if (c == '"') { size_t q = s.find('"', p + 1); if (q == std::string::npos) q = s.size() - 1; t.push_back(s.substr(p, q - p + 1)); p = q + 1; continue; }
            size_t q = p;
            while (q < s.size() && !std::isspace((unsigned char)s[q]) && s[q] != '(' && s[q] != ')' && s[q] != '{' && s[q] != '}' && s[q] != ';') q++;
            t.push_back(s.substr(p, q - p)); p = q;
        }
        return true;
    }
    bool done() const { return i >= t.size(); }
    // the next token as a label / scalar: false (with an error) when the file ends or the token is not a number
    bool label(long &v)
    {
        if (done()) { ffm_set_error("polyMesh: file ends where a label is expected"); return false; }
        const std::string &w = t[i]; char *end = nullptr; errno = 0;
        v = std::strtol(w.c_str(), &end, 10);
        if (end == w.c_str() || *end != 0 || errno) { ffm_set_error("polyMesh: '%s' is not a label", w.c_str()); return false; }
        i++; return true;
    }
    bool scalar(double &v)
    {
        if (done()) { ffm_set_error("polyMesh: file ends where a number is expected"); return false; }
        const std::string &w = t[i]; char *end = nullptr;
        v = std::strtod(w.c_str(), &end);
        if (end == w.c_str() || *end != 0) { ffm_set_error("polyMesh: '%s' is not a number", w.c_str()); return false; }
        i++; return true;
    }
    // a list size: 0 <= n and the file holds at least n more tokens (so a corrupt count cannot drive a huge allocation)
    bool count(long &n, size_t tokensPerItem)
    {
        if (!label(n)) return false;
        if (n < 0 || (size_t)n > (t.size() - i) / std::max<size_t>(tokensPerItem, 1)) { ffm_set_error("polyMesh: list size %ld does not fit the file", n); return false; }
        return true;
    }
    const std::string &peek() const { static const std::string e; return done() ? e : t[i]; }
    std::string next() { return done() ? std::string() : t[i++]; }
    bool expect(const char *s) { if (peek() == s) { i++; return true; } ffm_set_error("polyMesh: expected '%s', found '%s'", s, peek().c_str()); return false; }
    // FoamFile { ... }: returns false for binary format
    bool header()
    {
        if (peek() != "FoamFile") return true;
        i++; if (!expect("{")) return false;
        int depth = 1;
        while (!done() && depth > 0) {
            const std::string w = next();
            if (w == "{") depth++; else if (w == "}") depth--;
            else if (w == "format" && peek() != "ascii") { ffm_set_error("polyMesh: format %s is not supported (ascii only)", peek().c_str()); return false; }
        }
        return true;
    }
};

struct Patch { std::string name, type; int start = 0, size = 0; int myProcNo = -1, neighbProcNo = -1; };
}  // namespace

struct ffm_polymesh {
    std::vector<vec> points;
    std::vector<std::vector<int>> faces;
    std::vector<int> owner, neighbour;
    std::vector<Patch> patches;
    int nCells = 0;
    // geometry
    std::vector<vec> Cf, Sf, C;
    std::vector<double> V, weights, nonOrthDelta;
    std::vector<vec> corr;
};

static bool read_all(const std::string &dir, ffm_polymesh *m)
{
    {
        Tokens k; if (!k.load(dir + "/points") || !k.header()) return false;
        long n; if (!k.count(n, 5) || !k.expect("(")) return false;
        m->points.resize(n);
        for (long p = 0; p < n; p++) {
            if (!k.expect("(")) return false;
            for (int d = 0; d < 3; d++) if (!k.scalar(m->points[p][d])) return false;
            if (!k.expect(")")) return false;
        }
    }
    {
        Tokens k; if (!k.load(dir + "/faces") || !k.header()) return false;
        long n; if (!k.count(n, 6) || !k.expect("(")) return false;
        m->faces.resize(n);
        for (long f = 0; f < n; f++) {
            long nv; if (!k.count(nv, 1)) return false;
            if (nv < 3) { ffm_set_error("polyMesh: face %ld has %ld vertices", f, nv); return false; }
            if (!k.expect("(")) return false;
            m->faces[f].resize(nv);
            for (long v = 0; v < nv; v++) {
                long pt; if (!k.label(pt)) return false;
                if (pt < 0 || pt >= (long)m->points.size()) { ffm_set_error("polyMesh: face %ld refers to point %ld", f, pt); return false; }
                m->faces[f][v] = (int)pt;
            }
            if (!k.expect(")")) return false;
        }
    }
    for (int which = 0; which < 2; which++) {
        Tokens k; if (!k.load(dir + (which ? "/neighbour" : "/owner")) || !k.header()) return false;
        long n; if (!k.count(n, 1) || !k.expect("(")) return false;
        std::vector<int> &a = which ? m->neighbour : m->owner;
        a.resize(n);
        for (long f = 0; f < n; f++) {
            long c; if (!k.label(c)) return false;
            if (c < 0 || c > 0x7ffffffe) { ffm_set_error("polyMesh: %s of face %ld is %ld", which ? "neighbour" : "owner", f, c); return false; }
            a[f] = (int)c;
        }
    }
    if (m->owner.size() != m->faces.size() || m->neighbour.size() > m->faces.size()) { ffm_set_error("polyMesh: owner / neighbour / faces sizes disagree"); return false; }
    m->nCells = 0;
    for (int c : m->owner) m->nCells = std::max(m->nCells, c + 1);
    for (size_t f = 0; f < m->neighbour.size(); f++) {
        if (m->neighbour[f] <= m->owner[f]) { ffm_set_error("polyMesh: internal face %zu is not in upper-triangular order", f); return false; }
        m->nCells = std::max(m->nCells, m->neighbour[f] + 1);
    }
    // every cell owns or neighbours at least one face, so a valid mesh has fewer cells than 2 x faces
    if ((size_t)m->nCells > 2 * m->faces.size() + 1) { ffm_set_error("polyMesh: cell labels up to %d with %zu faces", m->nCells - 1, m->faces.size()); return false; }
    {
        Tokens k; if (!k.load(dir + "/boundary") || !k.header()) return false;
        long n; if (!k.count(n, 3) || !k.expect("(")) return false;
        for (long p = 0; p < n; p++) {
            Patch P; P.name = k.next(); if (!k.expect("{")) return false;
            int depth = 1;
            while (!k.done() && depth > 0) {
                const std::string w = k.next();
                if (w == "{") depth++; else if (w == "}") depth--;
                else if (depth == 1 && w == "type") P.type = k.next();
                else if (depth == 1 && w == "nFaces") { long v; if (!k.label(v) || v < 0 || v > (long)m->faces.size()) return false; P.size = (int)v; }
                else if (depth == 1 && w == "startFace") { long v; if (!k.label(v) || v < 0 || v > (long)m->faces.size()) return false; P.start = (int)v; }
                // processorPolyPatch of a processorN/constant/polyMesh/boundary written by decomposePar
                else if (depth == 1 && w == "myProcNo") { long v; if (!k.label(v) || v < 0) return false; P.myProcNo = (int)v; }
                else if (depth == 1 && w == "neighbProcNo") { long v; if (!k.label(v) || v < 0) return false; P.neighbProcNo = (int)v; }
            }
            if (P.start < (int)m->neighbour.size() || (long)P.start + P.size > (long)m->faces.size()) { ffm_set_error("polyMesh: patch %s out of range", P.name.c_str()); return false; }
            m->patches.push_back(P);
        }
    }
    return true;
}

static void geometry(ffm_polymesh *m)
{
    const size_t nF = m->faces.size(), nI = m->neighbour.size();
    m->Cf.resize(nF); m->Sf.resize(nF);
    for (size_t f = 0; f < nF; f++) {
        const std::vector<int> &v = m->faces[f]; const int n = (int)v.size();
        if (n == 3) {
            const vec &a = m->points[v[0]], &b = m->points[v[1]], &c = m->points[v[2]];
            m->Cf[f] = (1.0 / 3.0) * (a + b + c); m->Sf[f] = 0.5 * cross(b - a, c - a);
            continue;
        }
        vec fc = {0, 0, 0};
        for (int i = 0; i < n; i++) fc = fc + m->points[v[i]];
        fc = (1.0 / n) * fc;
        vec sumN = {0, 0, 0}, sumAc = {0, 0, 0}; double sumA = 0;
        for (int i = 0; i < n; i++) {
            const vec &p = m->points[v[i]], &q = m->points[v[(i + 1) % n]];
            const vec c = p + q + fc, nn = cross(q - p, fc - p); const double a = mag(nn);
            sumN = sumN + nn; sumA += a; sumAc = sumAc + a * c;
        }
        m->Cf[f] = sumA > 1e-300 ? (1.0 / 3.0) * ((1.0 / sumA) * sumAc) : fc;
        m->Sf[f] = 0.5 * sumN;
    }
    const int N = m->nCells;
    std::vector<vec> cEst(N, vec{0, 0, 0}); std::vector<int> nCellFaces(N, 0);
    for (size_t f = 0; f < nF; f++) { cEst[m->owner[f]] = cEst[m->owner[f]] + m->Cf[f]; nCellFaces[m->owner[f]]++; }
    for (size_t f = 0; f < nI; f++) { cEst[m->neighbour[f]] = cEst[m->neighbour[f]] + m->Cf[f]; nCellFaces[m->neighbour[f]]++; }
    for (int c = 0; c < N; c++) cEst[c] = (1.0 / std::max(nCellFaces[c], 1)) * cEst[c];
    m->C.assign(N, vec{0, 0, 0}); m->V.assign(N, 0.0);
    for (size_t f = 0; f < nF; f++) {
        const int o = m->owner[f];
        const double pv = dot(m->Sf[f], m->Cf[f] - cEst[o]);
        m->C[o] = m->C[o] + pv * (0.75 * m->Cf[f] + 0.25 * cEst[o]); m->V[o] += pv;
    }
    for (size_t f = 0; f < nI; f++) {
        const int n = m->neighbour[f];
        const double pv = dot(m->Sf[f], cEst[n] - m->Cf[f]);
        m->C[n] = m->C[n] + pv * (0.75 * m->Cf[f] + 0.25 * cEst[n]); m->V[n] += pv;
    }
    for (int c = 0; c < N; c++) { m->C[c] = std::fabs(m->V[c]) > 1e-300 ? (1.0 / m->V[c]) * m->C[c] : cEst[c]; m->V[c] *= 1.0 / 3.0; }
    m->weights.resize(nI); m->nonOrthDelta.resize(nI); m->corr.resize(nI);
    for (size_t f = 0; f < nI; f++) {
        const vec &Co = m->C[m->owner[f]], &Cn = m->C[m->neighbour[f]];
        const double own = std::fabs(dot(m->Sf[f], m->Cf[f] - Co)), nei = std::fabs(dot(m->Sf[f], Cn - m->Cf[f]));
        m->weights[f] = nei / (own + nei);
        const vec d = Cn - Co; const double ms = mag(m->Sf[f]); const vec nf = (1.0 / ms) * m->Sf[f];
        m->nonOrthDelta[f] = 1.0 / std::max(dot(nf, d), 0.05 * mag(d));
        m->corr[f] = nf - m->nonOrthDelta[f] * d;
    }
}

extern "C" int ffm_polymesh_read(const char *polyMeshDir, ffm_polymesh **out)
{
    if (!polyMeshDir || !out) return FFM_ERR_ARG;
    // nothing may leave an extern "C" function by exception (a file too large for memory, a corrupt count)
    ffm_polymesh *m = nullptr;
    try {
        m = new ffm_polymesh();
        if (!read_all(polyMeshDir, m)) { delete m; return FFM_ERR_ARG; }
        geometry(m);
    } catch (const std::exception &e) {
        ffm_set_error("polyMesh: %s", e.what());
        delete m;
        return FFM_ERR_ARG;
    }
    *out = m;
    return FFM_OK;
}
extern "C" int ffm_polymesh_destroy(ffm_polymesh *m) { delete m; return FFM_OK; }
extern "C" int ffm_polymesh_sizes(const ffm_polymesh *m, int *nPoints, int *nCells, int *nFaces, int *nInternalFaces, int *nPatches)
{
    if (!m) return FFM_ERR_ARG;
    if (nPoints) *nPoints = (int)m->points.size();
    if (nCells) *nCells = m->nCells;
    if (nFaces) *nFaces = (int)m->faces.size();
    if (nInternalFaces) *nInternalFaces = (int)m->neighbour.size();
    if (nPatches) *nPatches = (int)m->patches.size();
    return FFM_OK;
}
extern "C" int ffm_polymesh_addressing(const ffm_polymesh *m, int *lowerAddr, int *upperAddr)
{
    if (!m || !lowerAddr || !upperAddr) return FFM_ERR_ARG;
    for (size_t f = 0; f < m->neighbour.size(); f++) { lowerAddr[f] = m->owner[f]; upperAddr[f] = m->neighbour[f]; }
    return FFM_OK;
}
// cell arrays [N] / [3][N], internal-face arrays [F] / [3][F] (F = nInternalFaces); any pointer may be NULL
extern "C" int ffm_polymesh_geometry(const ffm_polymesh *m, double *V, double *C, double *Sf, double *Cf, double *magSf, double *weights,
                                     double *nonOrthDeltaCoeffs, double *nonOrthCorrectionVectors)
{
    if (!m) return FFM_ERR_ARG;
    const size_t N = m->nCells, F = m->neighbour.size();
    for (size_t c = 0; c < N; c++) { if (V) V[c] = m->V[c]; if (C) for (int d = 0; d < 3; d++) C[d * N + c] = m->C[c][d]; }
    for (size_t f = 0; f < F; f++) {
        for (int d = 0; d < 3; d++) { if (Sf) Sf[d * F + f] = m->Sf[f][d]; if (Cf) Cf[d * F + f] = m->Cf[f][d]; if (nonOrthCorrectionVectors) nonOrthCorrectionVectors[d * F + f] = m->corr[f][d]; }
        if (magSf) magSf[f] = mag(m->Sf[f]);
        if (weights) weights[f] = m->weights[f];
        if (nonOrthDeltaCoeffs) nonOrthDeltaCoeffs[f] = m->nonOrthDelta[f];
    }
    return FFM_OK;
}
// 1: patch i is a processor patch (`type processor`) and *myProcNo / *neighbProcNo are its entries; 0: it is not
extern "C" int ffm_polymesh_patch_processor(const ffm_polymesh *m, int i, int *myProcNo, int *neighbProcNo)
{
    if (!m || i < 0 || i >= (int)m->patches.size()) return FFM_ERR_ARG;
    const Patch &P = m->patches[i];
    if (P.type != "processor") return 0;
    if (P.myProcNo < 0 || P.neighbProcNo < 0) { ffm_set_error("polyMesh: processor patch %s without myProcNo / neighbProcNo", P.name.c_str()); return FFM_ERR_ARG; }
    if (myProcNo) *myProcNo = P.myProcNo;
    if (neighbProcNo) *neighbProcNo = P.neighbProcNo;
    return 1;
}
extern "C" int ffm_polymesh_patch(const ffm_polymesh *m, int i, char *name64, char *type32, int *startFace, int *nFaces)
{
    if (!m || i < 0 || i >= (int)m->patches.size()) return FFM_ERR_ARG;
    const Patch &P = m->patches[i];
    if (name64) { std::strncpy(name64, P.name.c_str(), 63); name64[63] = 0; }
    if (type32) { std::strncpy(type32, P.type.c_str(), 31); type32[31] = 0; }
    if (startFace) *startFace = P.start;
    if (nFaces) *nFaces = P.size;
    return FFM_OK;
}
// faceCells [n], Sf / Cf [3][n], deltaCoeffs [n] of patch i
extern "C" int ffm_polymesh_patch_geometry(const ffm_polymesh *m, int i, int *faceCells, double *Sf, double *Cf, double *deltaCoeffs)
{
    if (!m || i < 0 || i >= (int)m->patches.size()) return FFM_ERR_ARG;
    const Patch &P = m->patches[i]; const size_t n = P.size;
    for (size_t k = 0; k < n; k++) {
        const size_t f = P.start + k; const int c = m->owner[f];
        if (faceCells) faceCells[k] = c;
        for (int d = 0; d < 3; d++) { if (Sf) Sf[d * n + k] = m->Sf[f][d]; if (Cf) Cf[d * n + k] = m->Cf[f][d]; }
        if (deltaCoeffs) { const double ms = mag(m->Sf[f]); deltaCoeffs[k] = 1.0 / dot((1.0 / ms) * m->Sf[f], m->Cf[f] - m->C[c]); }
    }
    return FFM_OK;
}
