"""hypergef_amd: MI355X-native backend for the fused vertex -> hyperedge -> vertex
hypergraph aggregation (HGNNConv / UniGNNConv), behind the reference's operator
surface.  Kernels: hypergef_amd/csrc (HIP, gfx950); boundary: include/hg_aggr.h.
"""
import sys
import types

from . import _lib  # noqa: F401
from . import ops, plan, synth  # noqa: F401
from .balancer import balance_schedule  # noqa: F401
from .hypergraph import HyperGraph  # noqa: F401
from .ops import (HGNNAggr, HGNNAggrLinear, UniGNNConv, UniGNNConvdeg, UniGNNConvLinear,  # noqa: F401
                  hgnnaggr_linear)
from .plan import Plan  # noqa: F401

# The reference builds two top-level extension modules (setup.py:18,32-33).  They ship as real
# files next to this package (`hgnnaggr.py`, `unignnaggr.py`: repo root, or site-packages after
# `pip install .`), so `import hgnnaggr` needs no preparation.  The objects below are the same
# surface for an embedder that has only this package directory on its path.
hgnnaggr = types.ModuleType("hgnnaggr", "hgnnaggr (MI355X backend)")
for _n in ("hgnnaggr", "hgnnaggr_mean", "hgnnaggr_max"):
    setattr(hgnnaggr, _n, getattr(ops, _n))
unignnaggr = types.ModuleType("unignnaggr", "unignnaggr (MI355X backend)")
for _n in ("unignnaggrdeg", "unignnaggr", "unignnconvdeg", "unignnconv"):
    setattr(unignnaggr, _n, getattr(ops, _n))


def install_dropin():
    """Only needed when the top-level `hgnnaggr.py` / `unignnaggr.py` files are not on the import
    path: registers equivalent modules in sys.modules so the reference's wrappers
    (source/python/hgnnaggr.py:3, unignnconv.py:3) run unchanged.  Modules already importable
    (or imported) are left alone."""
    import importlib.util
    for name, mod in (("hgnnaggr", hgnnaggr), ("unignnaggr", unignnaggr)):
        if name in sys.modules:
            continue
        try:
            found = importlib.util.find_spec(name) is not None
        except (ImportError, ValueError):
            found = False
        if not found:
            sys.modules[name] = mod
