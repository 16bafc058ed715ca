"""One process per particle shard over the product driver: `smcmi_run_sharded` (csrc/run2.hpp `run2_impl`, csrc/sharded.hpp).

Replaces the reference's only parallel mode - `@distributed` over particles with the whole cloud serialised to every worker each
stage (src/smc_main.jl:169-170, 472-476, src/resample.jl:33-35) - by resident shards.  This module is only the plumbing around the
C ABI that `bench.py --gpus N` and the multi-process tests share:

  open_shard()   the Engine holding rank r's contiguous shard (global particle ids in the RNG: results do not depend on the
                 number of shards, bit for bit, for n_para <= 10)
  connect()      the communicator: RCCL over xGMI (`comm="rccl"`, one GPU per rank; unique id broadcast through torch.distributed)
                 or the host-mediated one (`comm="host"`: the library's collectives carried by torch.distributed on CPU tensors -
                 gloo - for ranks that share a GPU or have no RCCL)
  run()          smcmi_run_sharded - device likelihood families and host closures alike (a closure is registered on the shard's handle
                 and scores the proposals of the particles that shard holds: the reference's `parallel = true`, src/smc_main.jl:472-476)

The peer mailbox (include/smcmi.h) is set up by the library on the first run through whichever communicator is connected.
"""
from .engine import Engine, comm_unique_id, torch_dist_host_comm


def open_shard(spec, n_parts, n_para, rank, world, seed=0, device=0, max_stages=300, store_history=True):
    """Engine for shard `rank` of `world` equal contiguous shards with the model set; draw or upload the shard's cloud next."""
    if n_parts % world:
        raise ValueError("n_parts = %d is not divisible by the number of shards %d (equal contiguous shards)" % (n_parts, world))
    n_local = n_parts // world
    eng = Engine(n_parts, n_para, seed=seed, device=device, max_stages=max_stages, store_history=store_history, n_local=n_local,
                 gid0=rank * n_local)
    eng.set_model(spec)
    return eng


def connect(eng, rank, world, comm="rccl", group=None):
    """Join the sharded run's communicator; torch.distributed must be initialised (any backend for "rccl": it only broadcasts the
    128-byte id; a backend that moves CPU tensors for "host")."""
    import torch.distributed as dist

    if comm == "rccl":
        uid = [comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0, group=group)
        eng.comm_init(rank, world, uid[0])
    elif comm == "host":
        eng.comm_init_host(rank, world, *torch_dist_host_comm(group))
    else:
        raise ValueError("comm must be 'rccl' or 'host'")
    return eng


def run(eng, loglikelihood=None, old_loglikelihood=None, spec=None, **kw):
    """The sharded loop on rank's shard `eng`: smcmi_run_sharded (the product driver, communicator from connect()).  Host closures
    (`loglikelihood(theta (m, d)) -> (m,)`, the reference's user function: every worker scores the particles it holds) are registered
    on the handle first - open the shard with a model spec whose likelihood entry is ("host_callback", [], None, None); the driver then
    runs propose -> closure -> accept per MH step and block on every shard, everything else as for device families.  Closures passed here
    are registered AFTER the shard was opened: a run whose initial draw (`init_from_prior`) must score the closure registers it on the handle
    before that draw (`eng.set_likelihood_callback`) and calls run() without it.  `spec` (rounds 2-4 took the model here) is accepted and
    ignored: the model is the one open_shard() was given."""
    if spec is not None:
        import warnings
        warnings.warn("distributed.run(spec=...) is ignored: the model is set by open_shard()", DeprecationWarning, stacklevel=2)
    if loglikelihood is not None:
        eng.set_likelihood_callback(loglikelihood, which=0)
        if old_loglikelihood is not None:
            eng.set_likelihood_callback(old_loglikelihood, which=1)
    return eng.run_sharded(**kw)
