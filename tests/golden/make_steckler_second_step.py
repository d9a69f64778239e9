"""Adds the SECOND time step of the reference's golden log to tests/golden/steckler_first_step.json (key "second_step"): the Courant
numbers and deltaT printed in front of it, every solver line (field, solver, initial / final residual as printed, iterations), the
species table, min/max(T), the radiant fraction and the continuity errors -- cases/steckler/original/linux64/log.fireFoam:235-263,
read where it lies.  Data only.  Run from the repository root:  python tests/golden/make_steckler_second_step.py"""
import json
import os
import re

LOG = "/root/reference/cases/steckler/original/linux64/log.fireFoam"
HERE = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    lines = open(LOG).read().splitlines()
    i0 = next(i for i, ln in enumerate(lines) if ln.startswith("Time = 0.16"))
    i1 = next(i for i, ln in enumerate(lines) if ln.startswith("Time = 0.253333"))
    pre, body = lines[i0 - 4:i0], lines[i0:i1]
    co = next(mt for mt in (re.match(r"Courant Number mean: (\S+) max: (\S+)", ln) for ln in pre) if mt)
    dt = next(float(ln.split("=")[1]) for ln in pre if ln.startswith("deltaT"))
    out = {"source": "log.fireFoam:235-263", "courantMean": float(co.group(1)), "courantMax": float(co.group(2)),
           "deltaT": dt, "solves": [], "species_min_ave_max": {}, "continuity_errors": []}
    pat = re.compile(r"^(\w+):  Solving for (\w+), Initial residual = (\S+), Final residual = (\S+), No Iterations (\d+)$")
    for ln in body:
        mt = pat.match(ln)
        if mt:
            out["solves"].append({"name": mt.group(2), "solver": mt.group(1), "initialResidual": float(mt.group(3)), "finalResidual": float(mt.group(4)),
                                  "nIterations": int(mt.group(5))})
        mt = re.match(r"^\s*(\w+)\s+min/ave/max\s+=\s+(\S+)\s+(\S+)\s+(\S+)\s*$", ln)
        if mt:
            out["species_min_ave_max"][mt.group(1)] = [float(mt.group(k)) for k in (2, 3, 4)]
        if ln.startswith("Radiant Fraction is"):
            out["radiantFraction"] = float(ln.split()[-1])
        mt = re.match(r"min/max\(T\) = (\S+), (\S+)", ln)
        if mt:
            out["minmaxT"] = [float(mt.group(1)), float(mt.group(2))]
        mt = re.match(r"time step continuity errors : sum local = (\S+), global = (\S+), cumulative", ln)
        if mt:
            out["continuity_errors"].append({"sumLocal": float(mt.group(1)), "global": float(mt.group(2))})
    path = os.path.join(HERE, "steckler_first_step.json")
    d = json.load(open(path))
    d["second_step"] = out
    json.dump(d, open(path, "w"), indent=1)
    print("second step:", len(out["solves"]), "solves; deltaT", out["deltaT"])
