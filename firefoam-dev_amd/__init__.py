"""firefoam-dev_amd -- MI355X-native hot path of fireFoam (FV assembly + segregated sparse solves).

The directory name carries a hyphen, so import it through ``ffm_import.py`` at the
repository root (``from ffm_import import ffm``).  This package is a thin ctypes
binding of the C ABI in ``include/ffm.h`` (implemented by the HIP kernels in
``csrc/``) and mirrors the OpenFOAM interface names of the reference's hot path
(lduMatrix::Amul, lduMatrix::solver::solve, fvSolution keywords).  There is no CPU
fallback: every compute entry point needs libffm.so and a GPU.
"""
from .binding import (  # noqa: F401
    Context, lduMatrix, FfmError, lib, build, libpath, SOLVERS, PRECONDS, tile_hint_from_centres,
    renumber_levels, exported_symbols, declared_symbols, Plume, fvMesh, PyrolysisPanel, GAMG, Thermo,
)
from . import hexmesh  # noqa: F401
from . import decompose  # noqa: F401
from . import snippets  # noqa: F401  (host binding of the reference's equation files compiled over the Foam layer, where built)
from . import gloo_comm  # noqa: F401  (host transport over torch.distributed/gloo: several ranks on one GPU, CPU rehearsals)
