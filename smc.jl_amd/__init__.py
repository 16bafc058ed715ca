"""smc.jl_amd - MI355X-native SMC particle engine behind the FRBNY-DSGE/SMC.jl interface.

The directory name contains a dot, so import it as `smc_jl_amd` (repo-root shim smc_jl_amd.py).
Contents: csrc/ (HIP kernels + C ABI -> libsmcmi.so), host/ (ctypes binding and the Python mirror of the
reference's `smc(...)` / `Cloud` interface), julia/ (the ccall shim a Julia user loads).
"""
from .host import _lib  # noqa: F401
from .host.engine import Engine, comm_unique_id, run_group, torch_dist_host_comm  # noqa: F401
from .host.api import (Beta, CapmLiteral, LGSSKalman, Cloud, Gamma, GaussIso, InverseGamma, LinModel3, LinReg, Normal, Parameter,  # noqa: F401
                       RootInverseGamma, Uniform, cloud_isempty, get_accept, get_loglh, get_logpost, get_logprior,
                       get_old_loglh, get_vals, get_weights, parameter, smc, weighted_cov, weighted_mean, weighted_std, flatten_regimes, regime_values,
                       get_cloud, initial_draw, mutation, mvnormal_mixture_draw, resample)
from .host.cloudio import add_parameters_to_cloud, join_cloud, load_cloud, save_cloud, split_cloud  # noqa: F401

__all__ = ["Engine", "_lib", "smc", "Cloud", "parameter", "Normal", "Uniform", "Gamma", "Beta", "InverseGamma",
           "RootInverseGamma", "GaussIso", "LinReg", "LinModel3", "CapmLiteral", "LGSSKalman", "get_vals", "get_loglh", "get_logprior",
           "get_old_loglh", "get_logpost", "get_accept", "get_weights", "weighted_mean", "weighted_cov", "weighted_std",
           "cloud_isempty", "get_cloud", "initial_draw", "mutation", "mvnormal_mixture_draw", "resample", "split_cloud", "join_cloud", "add_parameters_to_cloud", "save_cloud", "load_cloud", "flatten_regimes", "regime_values"]
