"""Adds the 32 fvDOM ray solves of the first time step to tests/golden/steckler_first_step.json ("rays": name, iteration count,
initial and final residual as printed, "radiantFraction") from the reference's golden log where it lies:
cases/steckler/original/linux64/log.fireFoam:181-214 (`Radiation solver iter: 0`, `Radiant Fraction is 0.22`, 32 lines
`GAMG:  Solving for ILambda_<ray>_0, ...`).  Data only.  Run from the repository root:  python tests/golden/make_steckler_rays.py"""
import json
import os
import re

LOG = "/root/reference/cases/steckler/original/linux64/log.fireFoam"
HERE = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    lines = open(LOG).read().splitlines()[160:230]                 # the first time step
    pat = re.compile(r"^GAMG:  Solving for (ILambda_(\d+)_0), Initial residual = (\S+), Final residual = (\S+), No Iterations (\d+)$")
    rays = []
    for ln in lines:
        mt = pat.match(ln)
        if mt:
            assert int(mt.group(2)) == len(rays)
            rays.append({"name": mt.group(1), "solver": "GAMG", "initialResidual": float(mt.group(3)), "finalResidual": float(mt.group(4)),
                         "nIterations": int(mt.group(5))})
    assert len(rays) == 32
    frac = [float(ln.split()[-1]) for ln in lines if ln.startswith("Radiant Fraction is")]
    path = os.path.join(HERE, "steckler_first_step.json")
    d = json.load(open(path))
    d["rays"] = rays
    d["radiantFraction"] = frac[0]
    d["rays_source"] = "log.fireFoam:181-214 (fvDOM, nPhi 2, nTheta 4: 32 rays; Ii by GAMG + DILU to 1e-4, cases/steckler/system/fvSolution:63-73)"
    json.dump(d, open(path, "w"), indent=1)
    print("added", len(rays), "rays; radiant fraction", frac[0])
