"""Time the reference's equation files (examples/fireFoam_snippets.C -> libffm_refsnippets.so) on a box of n^3 cells next to the
compiled plume driver: the cost of the unfused class layer (one kernel and one temporary per operator).
usage: tests/probe_snippets.py n [steps]   (lives under tests/ because it takes its constants and mesh from the oracle)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from ffm_import import ffm
from oracle import plume            # constants and the mesh description only (probe script, not product)
from test_reference_snippets_gpu import SnippetCase

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = ffm.Context(0)
lib = C.CDLL(os.path.join(os.path.dirname(ffm.libpath()), "libffm_refsnippets.so"))
lib.firefoam_snippets_step.restype = C.c_int
lib.firefoam_snippets_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(SnippetCase)]
t0 = time.time()
gp = ffm.Plume(ctx, (n, n, n))                      # compiled driver: hydrostatic start state + the timing to compare with
m = plume.make_mesh((n, n, n), 0.05)
N, F = m.nCells, m.nFaces
B = sum(p.size for p in m.patches)
cOrd, fOrd = ffm.renumber_levels(N, m.l, m.u)
l2, u2, oldToNew = ffm.hexmesh.apply_renumbering(N, m.l, m.u, cOrd, fOrd)
A = ffm.lduMatrix(ctx, N, l2, u2)
patches = [(oldToNew[p.faceCells].astype(np.int32), p.Sf.T.copy(), p.deltaCoeffs) for p in m.patches]
mesh = ffm.fvMesh(A, m.V[cOrd], m.C[cOrd].T.copy(), m.Sf[fOrd].T.copy(), m.magSf[fOrd], m.weights[fOrd], m.deltaCoeffs[fOrd], patches)
mesh.set_face_centres(m.Cf[fOrd].T.copy())
print("set-up %.1f s, sweep mode %d" % (time.time() - t0, A.sweep_mode))
dp = C.POINTER(C.c_double); keep = []
h = lambda a: np.ascontiguousarray(a, np.float64)
def P(a):
    a = h(a); keep.append(a); return a.ctypes.data_as(dp)
def PP(arrs):
    arrs = [h(a) for a in arrs]; keep.append(arrs)
    arr = (dp * len(arrs))(*[a.ctypes.data_as(dp) for a in arrs]); keep.append(arr); return arr
bnd = lambda lst: np.concatenate(lst)
per = lambda fn: bnd([fn(p) for p in m.patches])
is_open = lambda p: p.name not in ("inlet", "floor")
fU = np.concatenate([per(lambda p, d=d: np.where(np.abs(p.Sf[:, d]) > 0, 0.0, -1.0) if is_open(p) else np.ones(p.size)) for d in range(3)])
refU = np.concatenate([per(lambda p, d=d: np.full(p.size, plume.U_IN if (p.name == "inlet" and d == 1) else 0.0)) for d in range(3)])
fixesU = per(lambda p: np.full(p.size, 0.0 if is_open(p) else 1.0))
fY = per(lambda p: np.full(p.size, 1.0 if p.name == "inlet" else (0.0 if p.name == "floor" else -1.0)))
refY = [per(lambda p, i=i: np.full(p.size, plume.Y_IN[i] if p.name == "inlet" else (plume.Y_AMB[i] if is_open(p) else 0.0))) for i in range(5)]
fH = per(lambda p: np.full(p.size, -1.0 if is_open(p) else 1.0))
refH = per(lambda p: np.full(p.size, plume.CP * (plume.T_IN - plume.TREF) if p.name == "inlet" else 0.0))
fluxMask = per(lambda p: np.full(p.size, 0.0 if is_open(p) else 1.0)); totalMask = 1.0 - fluxMask
G = plume.G; ghRef = -np.linalg.norm(G) * (n * 0.05)
gh = m.C @ G - ghRef; ghf = m.Cf @ G - ghRef; ghfb = bnd([p.Cf @ G - ghRef for p in m.patches])
st = {k: gp.field(k)[cOrd] for k in ("rho", "p", "p_rgh", "h", "K")}
st["U"] = np.stack([gp.field("U" + c)[cOrd] for c in "xyz"]); st["Y"] = [gp.field(s)[cOrd] for s in plume.SPECIES]
ph = gp.field("ph_rgh")
ph_b = bnd([np.zeros(p.size) if p.name == "top" else ph[p.faceCells] for p in m.patches])
st.update(dpdt=np.zeros(N), phi=np.zeros(F), phib=np.zeros(B), p_rghB=ph_b.copy())
os.environ["FFM_FOAM_QUIET"] = "1"
lib.firefoam_snippets_create.restype = C.c_void_p
lib.firefoam_snippets_create.argtypes = lib.firefoam_snippets_step.argtypes
lib.firefoam_snippets_advance.restype = C.c_int
lib.firefoam_snippets_advance.argtypes = [C.c_void_p, C.POINTER(SnippetCase), C.c_int]
lib.firefoam_snippets_destroy.argtypes = [C.c_void_p]
out = dict(rho=np.empty(N), U=np.empty((3, N)), p=np.empty(N), p_rgh=np.empty(N), h=np.empty(N), Y=[np.empty(N) for _ in range(5)],
           T=np.empty(N), K=np.empty(N), dpdt=np.empty(N), phi=np.empty(F), phib=np.empty(B), p_rghB=np.empty(B))
nit = (C.c_int * 32)()
yout = (dp * 5)(*[a.ctypes.data_as(dp) for a in out["Y"]])
cs = SnippetCase(deltaT=1e-3, RR=plume.RR, Cp=plume.CP, Tref=plume.TREF, pRef=plume.PREF, mu=plume.MU, Pr=plume.PR, sO2=plume.S_O2, HC=plume.HC,
    tau=plume.TAU, nSpecies=5, inertIndex=plume.INERT, fuelIndex=2, o2Index=0, W=P(plume.WMOL), nu=P(plume.NU),
    rho=P(st["rho"]), U=P(st["U"]), p=P(st["p"]), p_rgh=P(st["p_rgh"]), h=P(st["h"]), Y=PP(st["Y"]), K=P(st["K"]), dpdt=P(st["dpdt"]),
    phiF=P(st["phi"]), phiB=P(st["phib"]), gh=P(gh[cOrd]), ghfF=P(ghf[fOrd]), ghfB=P(ghfb), fU=P(fU), refU=P(refU), fixesU=P(fixesU),
    fY=P(fY), refY=PP(refY), fH=P(fH), refH=P(refH), fluxMaskP=P(fluxMask), totalMaskP=P(totalMask), ph_rgh_b=P(ph_b), p_rghB=P(st["p_rghB"]),
    rhoOut=out["rho"].ctypes.data_as(dp), UOut=out["U"].ctypes.data_as(dp), pOut=out["p"].ctypes.data_as(dp), p_rghOut=out["p_rgh"].ctypes.data_as(dp),
    hOut=out["h"].ctypes.data_as(dp), YOut=yout, TOut=out["T"].ctypes.data_as(dp), KOut=out["K"].ctypes.data_as(dp), dpdtOut=out["dpdt"].ctypes.data_as(dp),
    phiOutF=out["phi"].ctypes.data_as(dp), phiOutB=out["phib"].ctypes.data_as(dp), p_rghBOut=out["p_rghB"].ctypes.data_as(dp), nIterOut=nit, nIterCap=32)
solver = lib.firefoam_snippets_create(ctx.h, A.h, mesh.h, C.byref(cs))
for s in range(steps):
    ctx.sync(); t1 = time.perf_counter()
    k = lib.firefoam_snippets_advance(solver, C.byref(cs), 0)
    ctx.sync(); t2 = time.perf_counter()
    gp.step(); ctx.sync(); t3 = time.perf_counter()
    print("step %d: reference equation files on the device-resident state %.1f ms, compiled driver %.1f ms; iterations %s"
          % (s, 1e3 * (t2 - t1), 1e3 * (t3 - t2), list(nit[:k])))
k = lib.firefoam_snippets_advance(solver, C.byref(cs), 1); gp.step()
u_err = max(np.abs(out["U"][d] - gp.field("U" + "xyz"[d])[cOrd]).max() for d in range(3)) / max(np.abs(out["U"]).max(), 1e-30)
print("after %d steps: max|dU|/max|U| against the compiled driver %.1e" % (steps + 1, u_err))
lib.firefoam_snippets_destroy(solver)
