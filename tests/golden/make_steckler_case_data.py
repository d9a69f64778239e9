"""Writes tests/golden/steckler_case_data.json: the numeric case data of the reference's steckler case that the first-time-step
oracle needs (species table of constant/thermo.compressibleGas restricted to the species of constant/reactions, and the
single-step reaction), parsed from the case files where they lie (/root/reference/cases/steckler).  Data only: no text of the
reference is stored.  Run from the repository root:  python tests/golden/make_steckler_case_data.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import thermo as TH  # noqa: E402

CASE = "/root/reference/cases/steckler"


def build():
    tab = TH.parse_thermo_file(os.path.join(CASE, "constant/thermo.compressibleGas"))
    species, lhs, rhs = TH.parse_reaction(os.path.join(CASE, "constant/reactions"))
    table = {n: {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in tab[n].items()} for n in species}
    return {"source": "cases/steckler/constant/thermo.compressibleGas, cases/steckler/constant/reactions", "species": species,
            "table": table, "reaction": {"lhs": lhs, "rhs": rhs}, "fuel": "C3H8", "inertSpecie": "N2"}


if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "steckler_case_data.json")
    json.dump(build(), open(out, "w"), indent=1)
    print("wrote", out)
