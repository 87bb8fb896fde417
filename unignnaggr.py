"""Top-level module `unignnaggr`, as the reference installs it (setup.py:18,32-33;
HyperGsys/source/python/unignnconv.py:3 does `import unignnaggr`), on the MI355X backend
(unignnaggr.cc:81-102):

    unignnaggrdeg(balan_key, balan_row, group_st, group_ed, csrptr_t, indices_t, node_feat, degE, degV)
    unignnaggr(balan_key, balan_row, group_st, group_ed, csrptr_t, indices_t, node_feat)

plus `unignnconvdeg` / `unignnconv`, the names the reference's Python wrapper actually calls
(unignnconv.py:7,10) and its extension does not export (SURVEY.md defect D4).
"""
from hypergef_amd.ops import unignnaggr, unignnaggrdeg, unignnconv, unignnconvdeg  # noqa: F401

__all__ = ["unignnaggrdeg", "unignnaggr", "unignnconvdeg", "unignnconv"]
