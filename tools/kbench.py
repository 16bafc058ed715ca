#!/usr/bin/env python
"""Per-kernel timing of the stage kernels in isolation (HIP events, back-to-back launches) - development aid.
usage: python tools/kbench.py [n_parts ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smc_jl_amd import Engine  # noqa: E402
from tests import models  # noqa: E402

NAMES = ["pass16 p=0", "pass16 p=1 (+decide)", "correct (pass1 final)", "post_correct", "scan_weights", "resample_gather",
         "moments", "moments_reduce", "prepare_mutation", "mutate", "stage_begin", "empty"]
for n in [int(x) for x in sys.argv[1:]] or [100000]:
    e = Engine(n, 10, seed=1, max_stages=8, store_history=True)
    e.set_model(models.gauss_spec())
    e.init_from_prior()
    reps = 200 if n <= 1000000 else 30
    print("n_parts = %d" % n)
    for w, name in enumerate(NAMES):
        e.time_kernel(w, 5)
        print("  %-24s %9.2f us" % (name, e.time_kernel(w, reps)))
    e.close()
