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

# The reference builds two top-level extension modules (setup.py:18,32-33).
hgnnaggr = types.ModuleType("hgnnaggr", "hgnnaggr (MI355X backend)")
for _n in ("hgnnaggr", "hgnnaggr_mean", "hgnnaggr_max"):
    setattr(hgnnaggr, _n, getattr(ops, _n))
unignnaggr = types.ModuleType("unignnaggr", "unignnaggr (MI355X backend)")
for _n in ("unignnaggrdeg", "unignnaggr", "unignnconvdeg", "unignnconv"):
    setattr(unignnaggr, _n, getattr(ops, _n))


def install_dropin():
    """Make `import hgnnaggr` / `import unignnaggr` resolve to this backend, so the
    reference's wrappers (source/python/hgnnaggr.py:3, unignnconv.py:3) run unchanged."""
    sys.modules["hgnnaggr"] = hgnnaggr
    sys.modules["unignnaggr"] = unignnaggr
