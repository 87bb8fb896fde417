"""Top-level module `hgnnaggr`, as the reference installs it (setup.py:18,32-33 builds the
extension `hgnnaggr`; HyperGsys/source/python/hgnnaggr.py:3 does `import hgnnaggr`), on the
MI355X backend.  Same three functions, same positional signatures (hgnnaggr.cc:122-151):

    hgnnaggr(balan_key, balan_row, group_st, group_ed, csrptr_t, indices_t, node_feat, degE, degV, W)
    hgnnaggr_mean(csrptr_t, indices_t, node_feat, degE, degV, W)
    hgnnaggr_max(csrptr_t, indices_t, node_feat, degE, degV, W) -> [out, record_table]

A real file on the import path (repo root, or site-packages after `pip install .`): nothing has to
run before `import hgnnaggr` in a fresh interpreter.
"""
from hypergef_amd.ops import hgnnaggr, hgnnaggr_max, hgnnaggr_mean  # noqa: F401

__all__ = ["hgnnaggr", "hgnnaggr_mean", "hgnnaggr_max"]
