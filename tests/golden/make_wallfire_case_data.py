"""Writes tests/golden/wallfire_case_data.json: the model selections and numeric case data of the reference's wallFireSpread2D case
(BASELINE config 5) that the config-5 tests need -- the pyrolysis model and its coefficients, the solid's thermo / reaction / surface
radiation data, the schemes and boundary conditions of the panel region, the gas region's radiation model, the time controls -- parsed
from the case files where they lie (/root/reference/cases/wallFireSpread2D).  Data only: keywords and numbers, no text of the reference.
Run from the repository root:  python tests/golden/make_wallfire_case_data.py"""
import json
import os
import re

CASE = "/root/reference/cases/wallFireSpread2D"


def parse(path):
    """an OpenFOAM dictionary file as nested dicts; entries are token lists joined by blanks; lists in parentheses kept as strings"""
    txt = open(os.path.join(CASE, path)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    tok = re.findall(r'"[^"]*"|[{};]|[^\s{};()]+(?:\([^\s()]*\)[^\s{};()]*)*|\([^()]*(?:\([^()]*\)[^()]*)*\)', txt)
    pos = 0

    def block():
        nonlocal pos
        d = {}
        while pos < len(tok) and tok[pos] != "}":
            key = tok[pos].strip('"'); pos += 1
            if pos < len(tok) and tok[pos] == "{":
                pos += 1
                d[key] = block()
                pos += 1                                   # the closing brace
                if pos < len(tok) and tok[pos] == ";":
                    pos += 1
            else:
                val = []
                while pos < len(tok) and tok[pos] != ";":
                    if tok[pos] == "{":                    # e.g. `reactions { ... }` after a species list
                        pos += 1; val.append(block()); pos += 1
                        break
                    val.append(tok[pos]); pos += 1
                pos += 1
                d[key] = val[0] if len(val) == 1 else val
        return d
    return block()


def num(v):
    if isinstance(v, list):
        v = v[-1]
    return float(v)


def build():
    pz = parse("constant/pyrolysisZones")["pyrolysis"]
    model = pz["pyrolysisModel"]
    co = pz[model + "Coeffs"]
    rad = parse("constant/radiationProperties")
    dom = rad["fvDOMCoeffs"]
    ae = rad[rad["absorptionEmissionModel"] + "Coeffs"]
    srad = parse("constant/panelRegion/radiationProperties")
    sae = srad[srad["absorptionEmissionModel"] + "Coeffs"]
    th = parse("constant/panelRegion/thermo.solid")
    rx = parse("constant/panelRegion/reactions")
    reaction = list(rx["reactions"].values())[0] if isinstance(rx.get("reactions"), dict) else None
    if reaction is None:                                   # `reactions { ... }` parsed as the tail of the species list
        reaction = [v for v in rx.values() if isinstance(v, list) and isinstance(v[-1], dict)][0][-1]
        reaction = list(reaction.values())[0]
    order = float(re.match(r"\s*(\w+)\^([0-9.eE+-]+)", reaction["reaction"].strip('"')).group(2))
    sch = parse("system/panelRegion/fvSchemes")["laplacianSchemes"]
    T0 = parse("0/panelRegion/T")
    top = T0["boundaryField"]["panel_top"]
    cpl = T0["boundaryField"]["region0_to_panelRegion_panel"]
    ID = parse("0/IDefault")["boundaryField"]
    cd = parse("system/controlDict")
    ex = parse("system/extrudeToRegionMeshDict")
    U = parse("0/U")["boundaryField"]["region0_to_panelRegion_panel"]
    gsch = parse("system/fvSchemes")["divSchemes"]
    solid = lambda n: dict(rho=num(th[n]["equationOfState"]["rho"]), Cp=num(th[n]["thermodynamics"]["Cp"]), Hf=num(th[n]["thermodynamics"]["Hf"]),
                           kappa=num(th[n]["transport"]["kappa"]), absorptivity=num(sae[n]["absorptivity"]), emissivity=num(sae[n]["emissivity"]))
    return {
        "source": "cases/wallFireSpread2D: constant/{pyrolysisZones,radiationProperties}, constant/panelRegion/{radiationProperties,thermo.solid,reactions}, "
                  "system/panelRegion/fvSchemes, 0/panelRegion/T, 0/{IDefault,U}, system/{controlDict,extrudeToRegionMeshDict,fvSchemes}",
        "pyrolysis": {"pyrolysisModel": model, "gasHSource": co["gasHSource"], "qrHSource": co["qrHSource"], "moveMesh": co["moveMesh"],
                      "useChemistrySolvers": co["useChemistrySolvers"], "nLayers": int(num(ex["nLayers"])), "thickness": num(ex["linearNormalCoeffs"]["thickness"])},
        "solids": {"v": solid("v"), "char": solid("char")},
        "reaction": {"A": num(reaction["A"]), "Ta": num(reaction["Ta"]), "Tcrit": num(reaction["Tcrit"]), "order": order},
        "panelSchemes": {"laplacian(kappa,T)": sch["laplacian(kappa,T)"][1], "laplacian(thermo:alpha,h)": sch["laplacian(thermo:alpha,h)"][1]},
        "panelT": {"internalField": num(T0["internalField"]), "back": {"type": top["type"], "Tinf": num(top["Tinf"]), "h": num(top["h"])},
                   "coupled": {"type": cpl["type"], "neighbourFieldRadiativeName": cpl["neighbourFieldRadiativeName"], "emissivityMode": cpl["emissivityMode"]}},
        "hocSolid": num(U["hocSolid"]), "gasU": {"type": U["type"]},
        "radiation": {"radiationModel": rad["radiationModel"], "nPhi": int(num(dom["nPhi"])), "nTheta": int(num(dom["nTheta"])), "convergence": num(dom["convergence"]),
                      "maxIter": int(num(dom["maxIter"])), "solverFreq": int(num(rad["solverFreq"])), "absorptionEmissionModel": rad["absorptionEmissionModel"],
                      "Ehrr1": num(ae["Ehrr1"]), "Ehrr2": num(ae["Ehrr2"]), "radScaling": ae["radScaling"], "patch1": ae["patch1"], "patch2": ae["patch2"],
                      "div(Ji,Ii_h)": gsch["div(Ji,Ii_h)"][1],
                      "wallEmissivity": {k: (v["emissivityMode"], num(v["emissivity"]) if "emissivity" in v else None) for k, v in ID.items()}},
        "controls": {"deltaT": num(cd["deltaT"]), "maxCo": num(cd["maxCo"]), "maxDi": num(cd["maxDi"]), "maxDeltaT": num(cd["maxDeltaT"]), "adjustTimeStep": cd["adjustTimeStep"]},
    }


if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wallfire_case_data.json")
    json.dump(build(), open(out, "w"), indent=1)
    print("wrote", out)
