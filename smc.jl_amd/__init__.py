"""smc.jl_amd - MI355X-native SMC particle engine behind the FRBNY-DSGE/SMC.jl interface.

The directory name contains a dot, so import it as `smc_jl_amd` (repo-root shim smc_jl_amd.py).
Contents: csrc/ (HIP kernels + C ABI -> libsmcmi.so), host/ (ctypes binding and the Python mirror of the
reference's `smc(...)` / `Cloud` interface), julia/ (the ccall shim a Julia user loads).
"""
from .host import _lib  # noqa: F401
from .host.engine import Engine  # noqa: F401

__all__ = ["Engine", "_lib"]
