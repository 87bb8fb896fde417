"""Regenerates tests/golden/balancer_*.npz by RUNNING the reference's
HyperGsys/balancer.py (imported by file path; it needs only torch + numpy).

Run in the build container only (`python3 -B tests/golden/make_golden.py`);
/root/reference does not exist on the GPU box and nothing at test time reads it.
The .npz files hold data only: input csrptr + ngs and the four output lists.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF = "/root/reference/HyperGsys/balancer.py"


def load_reference_balancer():
    spec = importlib.util.spec_from_file_location("ref_balancer", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    from hypergef_amd import synth
    ref = load_reference_balancer()
    cases = {
        # name: (csrptr, ngs)
        "toy": (np.array([0, 3, 3, 8, 9], np.int32), 2),           # SURVEY.md 8(c) toy case
        "toy_trailing_empty": (np.array([0, 4, 9, 9, 9], np.int32), 3),
        "toy_leading_empty": (np.array([0, 0, 0, 5, 6], np.int32), 4),
        "exact_multiple": (np.array([0, 8, 16, 20], np.int32), 4),
        "cora_ngs210": (synth.cora_shape().csrptr, 210),            # hypergraph.py:74 table
        "citeseer_ngs6": (synth.citeseer_shape().csrptr, 6),
        "pubmed_ngs40": (synth.pubmed_shape().csrptr, 40),
        "ragged_ngs5": (synth.random_incidence(300, 200, 9.0, seed=7, empty_frac=0.1).csrptr, 5),
    }
    for name, (csrptr, ngs) in cases.items():
        bs = ref.balance_schedule(ngs, torch.from_numpy(np.asarray(csrptr, np.int32)))
        out = os.path.join(HERE, "balancer_%s.npz" % name)
        np.savez_compressed(out, csrptr=np.asarray(csrptr, np.int32), ngs=np.int32(ngs),
                            balan_key=np.asarray(bs.balan_key, np.int64),
                            balan_row=np.asarray(bs.balan_row, np.int64),
                            group_st=np.asarray(bs.group_st, np.int64),
                            group_ed=np.asarray(bs.group_ed, np.int64))
        print(name, len(bs.balan_key), len(bs.balan_row))


if __name__ == "__main__":
    main()
