"""CPU oracle for the run_contrack hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under contrack_amd/ may import this package.  Allowed users: tests/, the smoke test in
__graft_entry__.py, and bench.py's cpu_baseline leg.
"""
from .cpu_oracle import (build, lib, run_contrack, threshold_mask, label, np_sum,  # noqa: F401
                         row_weights, prepare_thresholds)
