"""Model specifications for the tests: the workloads of the package (smc.jl_amd/host/workloads.py) plus the bridge to the CPU oracle,
which only tests, smoke() and bench.py's cpu_baseline leg may touch."""
from smc_jl_amd.host.workloads import *  # noqa: F401,F403
from smc_jl_amd.host.workloads import KALMAN_KAPPA, KALMAN_TRUTH  # noqa: F401
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")        # the reference's fixtures (tests/golden/make_fixtures.py)


def oracle_model(spec):
    from oracle import oracle as orc

    return orc.model_from_spec(spec)
