"""include/ffmDictionary.H (the dictionary reader of the Foam layer) on the reference's own case files, read where they lie:
cases/steckler/system/{fvSolution,fvSchemes} and cases/wallFireSpread2D/system/{fvSolution,fvSchemes} -- regular-expression
keywords ("(Yi|h|k).*"), $macro expansion ($p_rgh; $U;), `Gauss multivariateSelection { ... }`, PIMPLE, relaxation factors.
Skipped where the reference is not mounted."""
import ctypes as C
import os

import numpy as np
import pytest

REF = "/root/reference/cases"
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "steckler", "system", "fvSolution")), reason="reference not mounted")
# ids of include/ffm.h (solver; preconditioner / smoother)
IDS = dict(PCG=0, PBICGSTAB=1, PBICG=2, DIAGONAL=3, SMOOTH=4, NONE=0, DIC=1, DILU=2, GS=3, SYMGS=4)


def _lib():
    import torch  # noqa: F401
    from ffm_import import ffm
    ffm.lib()
    lib = C.CDLL(os.path.join(os.path.dirname(ffm.libpath()), "libffm_b1demo.so"))
    lib.b1_read_case.restype = C.c_int
    lib.b1_read_case.argtypes = [C.c_char_p] * 5 + [C.POINTER(C.c_double), C.c_int]
    return lib, ffm


def _read(case, fields, schemes, multivariate):
    lib, ffm = _lib()
    out = np.zeros(256)
    n = lib.b1_read_case(os.path.join(REF, case, "system", "fvSolution").encode(), os.path.join(REF, case, "system", "fvSchemes").encode(),
                         " ".join(fields).encode(), " ".join(schemes).encode(), " ".join(multivariate).encode(),
                         out.ctypes.data_as(C.POINTER(C.c_double)), 256)
    assert 0 < n <= 256
    return out[:n], ffm


def test_steckler_dictionaries():
    fields = ["rho", "rhoFinal", "p_rgh", "p_rghFinal", "ph_rgh", "U", "UFinal", "Yi", "h", "kFinal", "Ii", "G"]
    schemes = ["div(phi,U)", "div(phi,K)", "div(Ji,Ii_h)", "div(phiU,p)", "div(((rho*nuEff)*dev2(T(grad(U)))))"]
    mv = ["div(phi,Yi_h):O2", "div(phi,Yi_h):C3H8", "div(phi,Yi_h):h"]
    out, ffm = _read("steckler", fields, schemes, mv)
    rows = out[:6 * len(fields)].reshape(len(fields), 6)
    r = dict(zip(fields, rows))
    assert tuple(r["p_rgh"][:4]) == (IDS["PCG"], IDS["DIC"], 1e-6, 0.01)
    assert tuple(r["p_rghFinal"][:4]) == (IDS["PCG"], IDS["DIC"], 1e-6, 0.0)          # $p_rgh; relTol 0.0;
    assert tuple(r["ph_rgh"][:4]) == (IDS["PCG"], IDS["DIC"], 1e-6, 0.01)             # $p_rgh;
    assert tuple(r["rhoFinal"][:4]) == (IDS["PCG"], IDS["DIC"], 1e-6, 0.0)            # "rho.*"
    for name in ("U", "UFinal"):                                                        # "U.*": smoothSolver symGaussSeidel maxIter 10
        assert tuple(r[name][:5]) == (IDS["SMOOTH"], IDS["SYMGS"], 1e-6, 0.0, 10)
    for name in ("Yi", "h", "kFinal"):                                                  # "(Yi|h|k).*": $U; tolerance 1e-8;
        assert tuple(r[name][:5]) == (IDS["SMOOTH"], IDS["SYMGS"], 1e-8, 0.0, 10)
    assert tuple(r["Ii"][:4]) == (5, IDS["DILU"], 1e-4, 0.0)                            # solver GAMG (FFM_GAMG = 5); smoother DILU
    assert tuple(r["G"][:2]) == (IDS["PCG"], IDS["DIC"])
    o = 6 * len(fields)
    assert tuple(out[o:o + 6]) == (1, 2, 0, 1, 1, 5)                                     # PIMPLE 1/2/0, hydrostaticInitialization yes, 5
    o += 6
    sch = out[o:o + 2 * len(schemes)].reshape(-1, 2)
    assert [tuple(x) for x in sch] == [(4, 1), (2, 1), (0, 1), (1, 1), (1, 1)]            # LUST, limitedLinear 1, upwind, linear, linear
    o += 2 * len(schemes)
    assert [tuple(x) for x in out[o:o + 6].reshape(-1, 2)] == [(3, 1), (3, 1), (2, 1)]   # O2, C3H8 limitedLinear01 1; h limitedLinear 1
    o += 6
    assert tuple(out[o:o + 3]) == (0, 0, -1)                                             # uncorrected, uncorrected, no equation factor


def test_wallFireSpread2D_dictionaries():
    out, _ = _read("wallFireSpread2D", ["p_rgh", "U", "Yi"], ["div(phi,U)"], [])
    assert tuple(out[:4]) == (5, 3, 1e-5, 0.01)           # p_rgh: solver GAMG; smoother GaussSeidel (FFM_GS = 3); tolerance 1e-5; relTol 0.01
    assert tuple(out[6 * 3 + 6:6 * 3 + 8]) == (6, 0.2)    # div(phi,U) Gauss filteredLinear2V 0.2 0.05: scheme 6, k = 0.2
    assert out[-3] == 1 and out[-2] == 1                  # Gauss linear corrected / corrected
    assert abs(out[-1] - 0.9) < 1e-12 or out[-1] == 1 or out[-1] == -1      # relaxationFactors::equations (0.9 in this case)


def _field(case, name, patches):
    lib, _ = _lib()
    lib.b1_read_field.restype = C.c_int
    lib.b1_read_field.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_double), C.c_int, C.c_char_p, C.c_int]
    out = np.zeros(1 + 5 * len(patches)); types = C.create_string_buffer(1024)
    n = lib.b1_read_field(os.path.join(REF, case, "0", name).encode(), " ".join(patches).encode(), out.ctypes.data_as(C.POINTER(C.c_double)), len(out), types, 1024)
    assert n == len(out)
    rows = out[1:].reshape(len(patches), 5)
    return out[0], dict(zip(patches, rows)), dict(zip(patches, types.value.decode().split()))


def test_steckler_field_files_hold_what_the_golden_log_match_rests_on():
    """The case data behind oracle/steckler.py, read from the reference's own 0/ files by include/ffmDictionary.H::fieldFile:
    ph_rgh is fixedValue 0 on `top` and fixedFluxPressure everywhere else; T = 298.15 K, O2 = 0.23301, N2 = 0.76699 inside;
    N2 is `calculated; value uniform 0` on top / sides / base / floor (the boundary mixture there is O2 alone until the first
    YEEqn -- the detail that makes the first final residual 0.0080439052); the baffle patches carry T = 300, O2 = 0.232,
    N2 = 0.768; the burner's velocity is (0 0 0) at t = 0."""
    P = ["top", "sides", "base", "burner", "floor", "baffle1DWall_master", "baffle1DWall_slave"]
    i, r, t = _field("steckler", "ph_rgh.orig", P)
    assert i == 0.0 and t["top"] == "fixedValue" and tuple(r["top"][:3]) == (1, 1, 0)          # known, f = 1, ref = 0
    assert all(t[p] == "fixedFluxPressure" and tuple(r[p][:4]) == (1, 0, 0, 0) for p in P[1:])
    i, r, t = _field("steckler", "T", P)
    assert i == 298.15 and r["baffle1DWall_master"][4] == 300 and r["baffle1DWall_slave"][4] == 300
    assert t["top"] == "inletOutlet" and tuple(r["top"][1:3]) == (-1, 298.15)                   # 1 - pos0(phi), inletValue
    i, r, t = _field("steckler", "O2", P)
    assert i == 0.23301 and t["baffle1DWall_master"] == "fixedValue" and r["baffle1DWall_master"][2] == 0.232
    i, r, t = _field("steckler", "N2", P)
    assert i == 0.76699 and r["baffle1DWall_slave"][2] == 0.768
    for p in ("top", "sides", "base", "floor"):
        assert t[p] == "calculated" and r[p][0] == 0 and r[p][4] == 0.0                          # no mixed form; value uniform 0
    i, r, t = _field("steckler", "U", ["burner", "base", "top"])
    assert i == 0.0 and t["burner"] == "flowRateInletVelocity" and r["burner"][4] == 0.0 and t["top"] == "pressureInletOutletVelocity"
    assert t["base"] == "noSlip"


def test_case_constants_of_the_oracle_come_from_the_reference_files():
    """the constants oracle/steckler.py, oracle/plume.py and bench.py quote from the reference's case, looked up in its files"""
    lib, _ = _lib()
    lib.b1_dict_lookup.restype = C.c_int
    lib.b1_dict_lookup.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]

    def look(rel, key):
        buf = C.create_string_buffer(512)
        assert lib.b1_dict_lookup(os.path.join(REF, "steckler", rel).encode(), key.encode(), buf, 512) >= 0, (rel, key)
        return buf.value.decode()
    assert look("constant/g", "value") == "(0 -9.81 0)" and float(look("constant/hRef", "value")) == 3.0
    assert float(look("constant/pRef", "value")) == 101325.0
    assert (look("constant/radiationProperties", "fvDOMCoeffs/nPhi"), look("constant/radiationProperties", "fvDOMCoeffs/nTheta")) == ("2", "4")
    assert look("constant/radiationProperties", "solverFreq") == "100" and look("constant/radiationProperties", "fvDOMCoeffs/maxIter") == "1"
    assert look("system/controlDict", "adjustTimeStep") == "yes" and float(look("system/controlDict", "maxCo")) == 0.9
    assert float(look("constant/thermo.compressibleGas", "O2/specie/molWeight")) == 31.9988
    assert float(look("constant/thermo.compressibleGas", "N2/specie/molWeight")) == 28.0134
    assert look("system/fvSolution", "solvers/UFinal/smoother") == "symGaussSeidel"      # through the pattern "U.*"
    assert look("system/fvSchemes", "divSchemes/div(Ji,Ii_h)") == "Gauss upwind"


def test_burner_patch_entries_of_the_steckler_case():
    """the two burner conditions' parameters read from the reference's own field files -- the mass-flow
    table of flowRateInletVelocity (cases/steckler/0/U:40-54: 0.03 kg/s at 0, 60, 100 s) and the massFluxFraction of
    totalFlowRateAdvectiveDiffusive (1 for the fuel, 0 for the other species: 0/C3H8:44-50, 0/O2, 0/Ydefault)"""
    lib, _ = _lib()
    lib.b1_read_function1.restype = C.c_int
    lib.b1_read_function1.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double)]
    xy = np.zeros(16); mff = C.c_double()
    n = lib.b1_read_function1(os.path.join(REF, "steckler", "0", "U").encode(), b"burner", b"massFlowRate", xy.ctypes.data_as(C.POINTER(C.c_double)), 16, None)
    assert n == 3 and xy[:6].tolist() == [0.0, 0.03, 60.0, 0.03, 100.0, 0.03]
    for name, want in (("C3H8", 1.0), ("O2", 0.0), ("Ydefault", 0.0), ("N2", 0.0)):
        lib.b1_read_function1(os.path.join(REF, "steckler", "0", name).encode(), b"burner", b"", None, 0, C.byref(mff))
        assert mff.value == want, name


@pytest.mark.parametrize("nCmpt", [1, 3])
def test_field_files_are_written_and_read_back(tmp_path, nCmpt):
    """SURVEY 8f N4, on-disk formats: include/ffmDictionary.H writes a volScalarField / volVectorField file the way OpenFOAM does
    with `writeFormat ascii; writePrecision 8` (cases/steckler/system/controlDict:36-38) -- banner, FoamFile header, `uniform v`
    for a constant list and `nonuniform List<scalar|vector> N ( ... )` otherwise -- and reads such a file back, element by
    element (8 significant digits).  The reference ships only `uniform` 0/ files, so the nonuniform form is pinned by OpenFOAM's
    documented layout (parity unpinned by reference data)."""
    lib, _ = _lib()
    dp = C.POINTER(C.c_double)
    nCells = 37
    rng = np.random.default_rng(11)
    internal = 300.0 + 50.0 * rng.standard_normal((nCells, nCmpt))
    names, types, sizes = ["top", "sides", "floor", "burner"], ["inletOutlet", "fixedValue", "zeroGradient", "fixedValue"], [5, 12, 7, 3]
    pv = [rng.standard_normal((n, nCmpt)) for n in sizes]
    pv[3][:] = pv[3][0]                                            # a constant list is written `uniform`
    pvals = np.ascontiguousarray(np.concatenate(pv))
    iout = np.zeros_like(internal); pout = np.full_like(pvals, np.nan)
    path = str(tmp_path / "T")
    lib.b1_field_roundtrip.restype = C.c_int
    lib.b1_field_roundtrip.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, dp, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), dp, dp, dp]
    nc = lib.b1_field_roundtrip(path.encode(), b"T", nCmpt, nCells, np.ascontiguousarray(internal).ctypes.data_as(dp), " ".join(names).encode(),
                                " ".join(types).encode(), (C.c_int * 4)(*sizes), pvals.ctypes.data_as(dp), iout.ctypes.data_as(dp), pout.ctypes.data_as(dp))
    assert nc == nCmpt
    assert np.abs(iout - internal).max() <= 5e-8 * np.abs(internal).max()
    off = 0
    for n, t, v in zip(sizes, types, pv):
        got = pout[off:off + n]
        if t == "zeroGradient":
            assert np.all(np.isnan(got))                               # no value entry written
        else:
            assert np.abs(got - v).max() <= 5e-8 * max(np.abs(v).max(), 1.0)
        off += n
    text = open(path).read()
    cls = "volScalarField" if nCmpt == 1 else "volVectorField"
    assert "class       %s;" % cls in text and 'location    "0.066666667";' in text and "object      T;" in text
    assert "internalField   nonuniform List<%s> \n%d\n(\n" % ("scalar" if nCmpt == 1 else "vector", nCells) in text
    first = ("%.8g" % internal[0, 0]) if nCmpt == 1 else "(" + " ".join("%.8g" % x for x in internal[0]) + ")"
    assert "(\n" + first + "\n" in text
    assert "    burner\n    {\n        type            fixedValue;\n        value           uniform " in text
    assert "        inletValue      uniform " in text and text.rstrip().endswith("// ************************************************************************* //")


def test_config5_selections_read_by_the_dictionary_reader_equal_the_fixture():
    """tests/golden/wallfire_case_data.json (the model selections the config-5 GPU tests run with; the GPU box has no case files) against
    the reference's own files read by include/ffmDictionary.H: the pyrolysis model and its coefficients, the panel region's schemes,
    solids, surface radiation and back-face condition, the gas region's fvDOM coefficients, emissivity modes and time controls."""
    import json
    lib, _ = _lib()
    lib.b1_dict_lookup.restype = C.c_int
    lib.b1_dict_lookup.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
    sel = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "wallfire_case_data.json")))

    def look(rel, key):
        buf = C.create_string_buffer(512)
        assert lib.b1_dict_lookup(os.path.join(REF, "wallFireSpread2D", rel).encode(), key.encode(), buf, 512) >= 0, (rel, key)
        return buf.value.decode()
    py = sel["pyrolysis"]
    model = look("constant/pyrolysisZones", "pyrolysis/pyrolysisModel")
    assert model == py["pyrolysisModel"] == "reactingOneDim21"
    for k in ("gasHSource", "qrHSource", "moveMesh", "useChemistrySolvers"):
        assert look("constant/pyrolysisZones", "pyrolysis/%sCoeffs/%s" % (model, k)) == py[k]
    assert int(look("system/extrudeToRegionMeshDict", "nLayers")) == py["nLayers"]
    assert float(look("system/extrudeToRegionMeshDict", "linearNormalCoeffs/thickness")) == py["thickness"]
    for name, d in sel["solids"].items():
        assert float(look("constant/panelRegion/thermo.solid", name + "/equationOfState/rho")) == d["rho"]
        assert float(look("constant/panelRegion/thermo.solid", name + "/thermodynamics/Cp")) == d["Cp"]
        assert float(look("constant/panelRegion/thermo.solid", name + "/thermodynamics/Hf")) == d["Hf"]
        assert float(look("constant/panelRegion/thermo.solid", name + "/transport/kappa")) == d["kappa"]
        assert float(look("constant/panelRegion/radiationProperties", "greyMeanSolidAbsorptionEmissionCoeffs/%s/emissivity" % name)) == d["emissivity"]
        assert float(look("constant/panelRegion/radiationProperties", "greyMeanSolidAbsorptionEmissionCoeffs/%s/absorptivity" % name)) == d["absorptivity"]
    for k, v in sel["panelSchemes"].items():
        assert look("system/panelRegion/fvSchemes", "laplacianSchemes/" + k).split()[:2] == ["Gauss", v]
    bk = sel["panelT"]["back"]
    assert look("0/panelRegion/T", "boundaryField/panel_top/type") == bk["type"]
    assert look("0/panelRegion/T", "boundaryField/panel_top/Tinf").split()[-1] == "293" and float(look("0/panelRegion/T", "boundaryField/panel_top/h").split()[-1]) == bk["h"]
    assert look("0/panelRegion/T", "boundaryField/region0_to_panelRegion_panel/emissivityMode") == sel["panelT"]["coupled"]["emissivityMode"]
    assert look("0/panelRegion/T", "boundaryField/region0_to_panelRegion_panel/neighbourFieldRadiativeName") == "qin"
    rd = sel["radiation"]
    assert look("constant/radiationProperties", "radiationModel") == rd["radiationModel"]
    for k in ("nPhi", "nTheta", "maxIter"):
        assert int(look("constant/radiationProperties", "fvDOMCoeffs/" + k)) == rd[k]
    assert float(look("constant/radiationProperties", "fvDOMCoeffs/convergence")) == rd["convergence"] and int(look("constant/radiationProperties", "solverFreq")) == rd["solverFreq"]
    assert look("constant/radiationProperties", "absorptionEmissionModel") == rd["absorptionEmissionModel"]
    for k in ("Ehrr1", "Ehrr2"):
        assert float(look("constant/radiationProperties", "constRadFractionEmissionCoeffs/" + k)) == rd[k]
    assert look("system/fvSchemes", "divSchemes/div(Ji,Ii_h)").split()[:2] == ["Gauss", rd["div(Ji,Ii_h)"]]
    assert look("0/IDefault", "boundaryField/region0_to_panelRegion_panel/emissivityMode") == "solidRadiation"
    assert float(look("0/U", "boundaryField/region0_to_panelRegion_panel/hocSolid")) == sel["hocSolid"]
    for k in ("maxCo", "maxDi", "maxDeltaT", "deltaT"):
        assert float(look("system/controlDict", k)) == sel["controls"][k]
