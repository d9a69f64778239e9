"""Writes tests/golden/steckler_log_steps.json: every time step of the reference's golden log
(cases/steckler/original/linux64/log.fireFoam, 29 steps, t = 0.0667 ... 2 s: from the cold start through ignition to a 1027 K flame),
read where it lies -- per step the Courant numbers and deltaT printed in front of it, every solver line (field, solver, initial and
final residual as printed, iterations; the 32 fvDOM ray solves of the first step are kept apart under "rays"), the species table,
min/max(T), the radiant fraction, the continuity errors.  Data only.  Run from the repository root:
    python tests/golden/make_steckler_log_steps.py"""
import json
import os
import re

LOG = "/root/reference/cases/steckler/original/linux64/log.fireFoam"
HERE = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    lines = open(LOG).read().splitlines()
    idx = [i for i, ln in enumerate(lines) if ln.startswith("Time = ")]
    pat = re.compile(r"^(\w+):  Solving for (\w+), Initial residual = (\S+), Final residual = (\S+), No Iterations (\d+)$")
    steps = []
    for a, b in zip(idx, idx[1:] + [len(lines)]):
        st = {"time": float(lines[a].split("=")[1]), "lines": [a + 1, b], "solves": [], "rays": [], "species_min_ave_max": {}, "continuity_errors": []}
        for ln in lines[a - 4:a]:
            mt = re.match(r"Courant Number mean: (\S+) max: (\S+)", ln)
            if mt:
                st["courantMean"], st["courantMax"] = float(mt.group(1)), float(mt.group(2))
            if ln.startswith("deltaT"):
                st["deltaT"] = float(ln.split("=")[1])
        for ln in lines[a:b]:
            mt = pat.match(ln)
            if mt:
                rec = {"name": mt.group(2), "solver": mt.group(1), "initialResidual": float(mt.group(3)), "finalResidual": float(mt.group(4)), "nIterations": int(mt.group(5))}
                (st["rays"] if rec["name"].startswith("ILambda_") else st["solves"]).append(rec)
            mt = re.match(r"^\s*(\w+)\s+min/ave/max\s+=\s+(\S+)\s+(\S+)\s+(\S+)\s*$", ln)
            if mt:
                st["species_min_ave_max"][mt.group(1)] = [float(mt.group(k)) for k in (2, 3, 4)]
            if ln.startswith("Radiant Fraction is"):
                st["radiantFraction"] = float(ln.split()[-1])
            mt = re.match(r"min/max\(T\) = (\S+), (\S+)", ln)
            if mt:
                st["minmaxT"] = [float(mt.group(1)), float(mt.group(2))]
            mt = re.match(r"time step continuity errors : sum local = (\S+), global = (\S+), cumulative", ln)
            if mt:
                st["continuity_errors"].append({"sumLocal": float(mt.group(1)), "global": float(mt.group(2))})
            if ln.startswith("ExecutionTime"):
                break
        steps.append(st)
    out = {"source": "reference cases/steckler/original/linux64/log.fireFoam (2017-08-28, OpenFOAM dev-tmp-0c4175bec707, nProcs 1): all time steps, numbers as printed",
           "steps": steps}
    json.dump(out, open(os.path.join(HERE, "steckler_log_steps.json"), "w"), indent=0)
    print(len(steps), "steps,", sum(len(s["solves"]) for s in steps), "solver lines")
