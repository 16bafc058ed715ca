"""Engine: thin object wrapper over one libsmcmi handle (= one particle shard on one MI355X).

Method names follow the reference functions they stand in for (src/helpers.jl, src/resample.jl,
src/particle.jl, src/mutation.jl); arrays are numpy float64, clouds are (N, R) Fortran-ordered like
the reference's `cloud.particles`.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, dp, ip, lp


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _d(a):
    return a.ctypes.data_as(dp)


def _i(a):
    return a.ctypes.data_as(ip)


class Engine:
    def __init__(self, n_parts, n_para, seed=0, device=0, max_stages=300, store_history=True, n_local=None, gid0=0):
        self._L = _lib.lib()
        self.n_parts, self.d, self.R = int(n_parts), int(n_para), int(n_para) + 5
        self.n = int(n_parts if n_local is None else n_local)
        self.gid0, self.seed, self.max_stages, self.store_history = int(gid0), int(seed), int(max_stages), bool(store_history)
        cfg = _lib.Config(self.n_parts, self.n, self.gid0, self.d, device, self.seed, self.max_stages, int(store_history))
        self._h = C.c_void_p()
        check(self._L.smcmi_create(C.byref(cfg), C.byref(self._h)))
        self.free_inds = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.smcmi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- model -------------------------------------------------------------------------------------
    def set_parameters(self, priors, bounds, fixed=None):
        d = self.d
        assert len(priors) == d and len(bounds) == d
        fam = np.array([_lib.PRIOR[p[0]] if isinstance(p[0], str) else int(p[0]) for p in priors], dtype=np.int32)
        a, b = _f64([p[1] for p in priors]), _f64([p[2] for p in priors])
        lo, hi = _f64([x[0] for x in bounds]), _f64([x[1] for x in bounds])
        fx = np.zeros(d, dtype=np.int32) if fixed is None else np.ascontiguousarray(fixed, dtype=np.int32)
        check(self._L.smcmi_set_parameters(self._h, _i(fx), _d(lo), _d(hi), _i(fam), _d(a), _d(b)))
        self.free_inds = np.flatnonzero(fx == 0).astype(np.int32)

    def set_likelihood(self, family, par=(), data=None, aux=None, which=0):
        fam = _lib.LIK[family] if isinstance(family, str) else int(family)
        par = _f64(par).ravel()
        data = None if data is None else np.asfortranarray(np.atleast_2d(np.asarray(data, dtype=np.float64)))
        aux = None if aux is None else np.asfortranarray(np.atleast_2d(np.asarray(aux, dtype=np.float64)))
        check(self._L.smcmi_set_likelihood(
            self._h, which, fam, _d(par) if par.size else None, par.size,
            None if data is None else _d(data), 0 if data is None else data.shape[0], 0 if data is None else data.shape[1],
            None if aux is None else _d(aux), 0 if aux is None else aux.shape[0], 0 if aux is None else aux.shape[1]))

    def set_likelihood_callback(self, fn, which=0):
        """Register a host likelihood (smcmi_set_likelihood_callback): fn(theta) with theta an (m, d) array of proposals that
        passed the bounds check, returning m log-likelihoods (-inf allowed).  fn = None unregisters.  The engine keeps the ctypes
        trampoline alive; exceptions inside fn abort the run (SMCMI_ERR_CALLBACK) and are re-raised by run()."""
        if not hasattr(self, "_cb"):
            self._cb, self._cb_exc = [None, None], None
        if fn is None:
            self._cb[which] = None
            check(self._L.smcmi_set_likelihood_callback(self._h, which, None, None))
            return

        def tramp(theta_p, m, d, out_p, _ud):
            try:
                th = np.ctypeslib.as_array(theta_p, shape=(d, m)).T          # column-major m x d
                out = np.ctypeslib.as_array(out_p, shape=(m,))
                out[:] = np.asarray(fn(th), dtype=np.float64).reshape(m)
                return 0
            except BaseException as ex:   # noqa: BLE001 - must not unwind through the C frames
                self._cb_exc = ex
                return 1

        cb = _lib.LIK_CALLBACK(tramp)
        self._cb[which] = cb
        check(self._L.smcmi_set_likelihood_callback(self._h, which, C.cast(cb, C.c_void_p), None))

    def eval_cloud_callback(self, which=0, column=None):
        self._checked(self._L.smcmi_eval_cloud_callback(self._h, which, self.d if column is None else int(column)))

    def callback_stats(self):
        a, b = C.c_int64(), C.c_int64()
        check(self._L.smcmi_callback_stats(self._h, C.byref(a), C.byref(b)))
        return dict(calls=a.value, evaluations=b.value)

    def callback_phases(self):
        """ms the last run with a host callback spent per phase on the calling thread (include/smcmi.h smcmi_callback_phases)"""
        out = (C.c_double * 8)()
        check(self._L.smcmi_callback_phases(self._h, out, 8))
        names = ("first_chunk_wait", "later_chunk_wait", "pack", "callback", "scatter", "enqueue", "stage_device_part")
        return {k: out[i] for i, k in enumerate(names)}

    def set_model(self, spec):
        """spec: dict(priors, bounds, fixed, lik=(family, par, data, aux), old_lik=None|(...))."""
        self.set_parameters(spec["priors"], spec["bounds"], spec.get("fixed"))
        lk = spec["lik"]
        self.set_likelihood(lk[0], lk[1], lk[2], lk[3], which=0)
        ol = spec.get("old_lik")
        if ol is None:
            self.set_likelihood("none", which=1)
        else:
            self.set_likelihood(ol[0], ol[1], ol[2], ol[3], which=1)

    # ---- cloud -------------------------------------------------------------------------------------
    def upload_cloud(self, particles):
        p = np.asfortranarray(np.asarray(particles, dtype=np.float64))
        assert p.shape == (self.n, self.R), (p.shape, (self.n, self.R))
        check(self._L.smcmi_upload_cloud(self._h, _d(p)))

    def upload_cloud_from_device(self, dev_ptr):
        """Device-to-device restore from a resident n x R column-major buffer (e.g. a torch tensor's data_ptr())."""
        check(self._L.smcmi_upload_cloud_device(self._h, C.c_void_p(int(dev_ptr))))

    def download_cloud(self):
        p = np.empty((self.n, self.R), order="F")
        check(self._L.smcmi_download_cloud(self._h, _d(p)))
        return p

    def init_from_prior(self):
        self._checked(self._L.smcmi_init_from_prior(self._h))

    def initialize_likelihoods(self):
        """initialize_likelihoods!: old_loglh <- loglh, then loglh / logprior on the (new) data."""
        self._checked(self._L.smcmi_initialize_likelihoods(self._h))

    # ---- stage primitives --------------------------------------------------------------------------
    def ess_at(self, phis, phi_prev):
        phis = _f64(np.atleast_1d(phis))
        out = np.empty_like(phis)
        check(self._L.smcmi_ess_at(self._h, _d(phis), phis.size, phi_prev, _d(out)))
        return out

    def solve_phi(self, sched, j, phi_prop, phi_prev, tempering_target, ess_prev, resampled_last):
        sched = _f64(sched)
        jj, pp, rl, out = C.c_int32(j), C.c_double(phi_prop), C.c_int32(int(resampled_last)), C.c_double()
        check(self._L.smcmi_solve_phi(self._h, _d(sched), sched.size, C.byref(jj), C.byref(pp), phi_prev, tempering_target,
                                      ess_prev, C.byref(rl), C.byref(out)))
        return out.value, bool(rl.value), jj.value, pp.value

    def correct(self, phi_n, phi_prev, prior_weight=0.0, log_prob_old_data=0.0, threshold_ratio=0.5):
        st = _lib.StageStats()
        check(self._L.smcmi_correct(self._h, phi_n, phi_prev, prior_weight, log_prob_old_data, threshold_ratio, C.byref(st)))
        return dict(ess=st.ess, sum_unnorm=st.sum_unnorm, logz_inc=st.logz_inc, resample=bool(st.resample))

    def resample(self, method="systematic", stage=0, offsets=None):
        anc = np.empty(self.n, dtype=np.int64)
        off = None if offsets is None else _f64(np.atleast_1d(offsets))
        check(self._L.smcmi_resample(self._h, _lib.RESAMPLE[method], stage, None if off is None else _d(off),
                                     anc.ctypes.data_as(lp)))
        return anc

    def bridge_resample_from(self, old, n_out, method="systematic", stage=0, offsets=None):
        """Rows [0, n_out) of this cloud <- resample(get_weights(old); n_parts = n_out) rows of `old`, weights included
        (src/smc_main.jl:266-279)."""
        anc = np.empty(max(int(n_out), 1), dtype=np.int64)
        off = None if offsets is None else _f64(np.atleast_1d(offsets))
        check(self._L.smcmi_bridge_resample(self._h, old._h, _lib.RESAMPLE[method], stage, int(n_out),
                                            None if off is None else _d(off), anc.ctypes.data_as(lp)))
        return anc[:int(n_out)]

    def copy_rows_from(self, src, n_rows, dst_row0=0, src_row0=0):
        check(self._L.smcmi_copy_rows(self._h, int(dst_row0), src._h, int(src_row0), int(n_rows)))

    def normalize_weights(self, zero_bad_loglh=False):
        """zero_bad_loglh_weights! (optional) + normalize_weights! (src/particle.jl:362-366, 392-396)."""
        check(self._L.smcmi_normalize_weights(self._h, int(zero_bad_loglh)))

    def moments(self):
        mean, cov = np.empty(self.d), np.empty((self.d, self.d))
        check(self._L.smcmi_moments(self._h, _d(mean), _d(cov)))
        return mean, cov

    def mutate(self, mu_free, Sigma_free, block_ptr, blocks_free, phi_n, phi_prev, c, alpha, n_mh_steps, stage):
        mu, S = _f64(mu_free), _f64(Sigma_free)
        bp, bf = np.ascontiguousarray(block_ptr, dtype=np.int32), np.ascontiguousarray(blocks_free, dtype=np.int32)
        acc = C.c_double()
        check(self._L.smcmi_mutate(self._h, _d(mu), _d(S), _i(bp), _i(bf), bp.size - 1, phi_n, phi_prev, c, alpha, n_mh_steps,
                                   stage, C.byref(acc)))
        return acc.value

    def propose(self, mu_free, Sigma_free, block_ptr, blocks_free, block, mh_step, c, alpha, stage):
        mu, S = _f64(mu_free), _f64(Sigma_free)
        bp, bf = np.ascontiguousarray(block_ptr, dtype=np.int32), np.ascontiguousarray(blocks_free, dtype=np.int32)
        prop = np.empty((self.n, self.d), order="F")
        lpr, qd = np.empty(self.n), np.empty(self.n)
        check(self._L.smcmi_propose(self._h, _d(mu), _d(S), _i(bp), _i(bf), bp.size - 1, block, mh_step, c, alpha, stage,
                                    _d(prop), _d(lpr), _d(qd)))
        return prop, lpr, qd

    def accept(self, loglik_new, loglik_old_new, phi_n, block, mh_step, n_blocks, stage, last):
        ln = _f64(loglik_new)
        lo = None if loglik_old_new is None else _f64(loglik_old_new)
        check(self._L.smcmi_accept(self._h, _d(ln), None if lo is None else _d(lo), phi_n, block, mh_step, n_blocks, stage,
                                   int(last)))

    # ---- whole loop --------------------------------------------------------------------------------
    def run(self, n_blocks=1, n_mh_steps=1, lam=2.1, n_phi=300, resampling_method="systematic", threshold_ratio=0.5,
            c=0.5, alpha=1.0, target=0.25, use_fixed_schedule=True, tempering_target=0.97, prior_weight=0.0,
            log_prob_old_data=0.0, solver_passes=0, sync_every=0, use_graph=0, phi_rtol=0.0, initial_ess=0.0,
            stop_after_stage=0, continue_run=False):
        """The whole loop on the device.  `stop_after_stage` = k pauses once cloud.stage_index has reached k (result["paused"]);
        `continue_run` goes on from the handle's loop state (after a pause or set_loop_state) - the device side of
        save_intermediate / continue_intermediate (src/smc_main.jl:334-361, 499-507)."""
        rc = self._run_config(n_blocks, n_mh_steps, lam, n_phi, resampling_method, threshold_ratio, c, alpha, target,
                              use_fixed_schedule, tempering_target, prior_weight, log_prob_old_data, solver_passes, sync_every,
                              use_graph, phi_rtol, initial_ess)
        rc.stop_after_stage, rc.continue_run = int(stop_after_stage), int(bool(continue_run))
        res = _lib.Result()
        self._checked(self._L.smcmi_run(self._h, C.byref(rc), C.byref(res)))
        out = self._result(res)
        out["paused"] = bool(res.paused)
        return out

    def _checked(self, rc):
        """check(rc); an exception raised inside a Python likelihood callback travels through the C frames as SMCMI_ERR_CALLBACK
        and is re-raised here."""
        exc = getattr(self, "_cb_exc", None)
        if rc != 0 and exc is not None:
            self._cb_exc = None
            raise exc
        check(rc)

    def get_loop_state(self):
        s = _lib.LoopState()
        check(self._L.smcmi_get_loop_state(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in _lib.LoopState._fields_}

    def set_loop_state(self, **fields):
        s = _lib.LoopState()
        for k, v in fields.items():
            setattr(s, k, v)
        check(self._L.smcmi_set_loop_state(self._h, C.byref(s)))

    def set_stage_records(self, schedule, ess, c_hist, accept_hist, resampled):
        a = [_f64(x) for x in (schedule, ess, c_hist, accept_hist)]
        rs = np.ascontiguousarray(resampled, dtype=np.int32)
        check(self._L.smcmi_set_stage_records(self._h, a[0].size, _d(a[0]), _d(a[1]), _d(a[2]), _d(a[3]), _i(rs)))

    def set_history(self, w, W):
        w, W = np.asfortranarray(w, dtype=np.float64), np.asfortranarray(W, dtype=np.float64)
        check(self._L.smcmi_set_history(self._h, w.shape[1], _d(w), _d(W)))

    def _run_config(self, n_blocks, n_mh_steps, lam, n_phi, resampling_method, threshold_ratio, c, alpha, target,
                    use_fixed_schedule, tempering_target, prior_weight, log_prob_old_data, solver_passes, sync_every, use_graph,
                    phi_rtol, initial_ess=0.0):
        return _lib.RunConfig(n_blocks, n_mh_steps, lam, n_phi, _lib.RESAMPLE[resampling_method], threshold_ratio, c, alpha,
                              target, int(use_fixed_schedule), tempering_target, prior_weight, log_prob_old_data, solver_passes,
                              sync_every, use_graph, initial_ess, phi_rtol)

    @staticmethod
    def _result(res):
        return dict(n_stages=res.n_stages, resamples=res.resamples, logmdd=res.logmdd, c=res.c, accept=res.accept,
                    seconds=res.seconds, kernel_ms_mutate=res.kernel_ms_mutate, n_mutate_launches=res.n_mutate_launches,
                    solver_passes=res.solver_passes, solver_stalls=res.solver_stalls, select_stalls=res.select_stalls, spec_stalls=res.spec_stalls,
                    n_segments=res.n_segments, segment_stages=res.segment_stages, kernel_ms_segments=res.kernel_ms_segments,
                    segment_blocks=res.segment_blocks, segment_state=res.segment_state, segment_timeouts=res.segment_timeouts,
                    shift_fallback_stage=res.shift_fallback_stage)

    # ---- sharded whole-loop drivers (csrc/sharded.hpp) ---------------------------------------------------
    def comm_init(self, rank, world, unique_id):
        """Join the RCCL communicator (unique_id: the 128 bytes of comm_unique_id() from rank 0)."""
        check(self._L.smcmi_comm_init(self._h, rank, world, bytes(unique_id)))

    def comm_init_host(self, rank, world, allgather, alltoallv=None, barrier=None):
        """Host-mediated communicator (include/smcmi.h: smcmi_comm_init_host) from Python callables on numpy arrays:
        allgather(send[count]) -> array[world * count] in rank order; alltoallv(list of per-peer send arrays) -> list of per-peer
        receive arrays (recv_counts given as second argument); barrier().  `torch_dist_host_comm()` builds the three from
        torch.distributed (gloo).  Exceptions inside the callables abort the collective (SMCMI_ERR_CALLBACK) and are re-raised."""
        world = int(world)

        def guard(fn):
            def g(*a):
                try:
                    fn(*a)
                    return 0
                except BaseException as ex:      # noqa: BLE001 - must not unwind through the C frames
                    self._cb_exc = ex
                    return 1
            return g

        def c_allgather(send_p, recv_p, count, _ud):
            send = np.ctypeslib.as_array(send_p, shape=(int(count),))
            out = np.ctypeslib.as_array(recv_p, shape=(world * int(count),))
            out[:] = np.asarray(allgather(send.copy()), dtype=np.float64).reshape(-1)

        def c_alltoallv(send_p, sc_p, sd_p, recv_p, rc_p, rd_p, _ud):
            sc, sd = np.ctypeslib.as_array(sc_p, shape=(world,)), np.ctypeslib.as_array(sd_p, shape=(world,))
            rcn, rd = np.ctypeslib.as_array(rc_p, shape=(world,)), np.ctypeslib.as_array(rd_p, shape=(world,))
            stot, rtot = int(sc.sum()), int(rcn.sum())
            send = np.ctypeslib.as_array(send_p, shape=(max(stot, 1),))
            recv = np.ctypeslib.as_array(recv_p, shape=(max(rtot, 1),))
            got = alltoallv([send[int(sd[p]):int(sd[p] + sc[p])].copy() for p in range(world)], [int(x) for x in rcn])
            for p in range(world):
                if rcn[p]:
                    recv[int(rd[p]):int(rd[p] + rcn[p])] = np.asarray(got[p], dtype=np.float64).reshape(-1)

        def c_barrier(_ud):
            if barrier is not None:
                barrier()

        if not hasattr(self, "_cb_exc"):
            self._cb_exc = None
        hc = _lib.HostComm()
        hc.allgather = _lib.HC_ALLGATHER(guard(c_allgather))
        hc.alltoallv = _lib.HC_ALLTOALLV(guard(c_alltoallv)) if alltoallv is not None else C.cast(None, _lib.HC_ALLTOALLV)
        hc.barrier = _lib.HC_BARRIER(guard(c_barrier))
        hc.user = None
        self._hostc = hc                             # keep the trampolines alive as long as the handle
        self._checked(self._L.smcmi_comm_init_host(self._h, int(rank), world, C.byref(hc)))

    def mailbox_export(self):
        """64-byte IPC handle of this handle's peer-mailbox table (include/smcmi.h: smcmi_mailbox_export)."""
        buf = (C.c_uint8 * 64)()
        self._checked(self._L.smcmi_mailbox_export(self._h, buf))
        return bytes(buf)

    def mailbox_import(self, rank, world, handles):
        """Map every rank's table (handles: rank-ordered list of 64-byte handles)."""
        raw = b"".join(handles)
        buf = (C.c_uint8 * len(raw)).from_buffer_copy(raw)
        self._checked(self._L.smcmi_mailbox_import(self._h, int(rank), int(world), buf))

    def mailbox_selftest(self, rank, world, rounds=256):
        """Mismatches + time-outs of `rounds` test exchanges with every peer (all ranks call it together)."""
        errs = C.c_int32(0)
        self._checked(self._L.smcmi_mailbox_selftest(self._h, int(rank), int(world), int(rounds), C.byref(errs)))
        return int(errs.value)

    def mailbox_active(self):
        """True if the last sharded run handed its per-stage sums over through the peer mailbox (else: all-gathers)."""
        a = C.c_int32(0)
        self._checked(self._L.smcmi_mailbox_active(self._h, C.byref(a)))
        return bool(a.value)

    def run_sharded(self, n_blocks=1, n_mh_steps=1, lam=2.1, n_phi=300, resampling_method="systematic", threshold_ratio=0.5,
                    c=0.5, alpha=1.0, target=0.25, use_fixed_schedule=True, tempering_target=0.97, prior_weight=0.0,
                    log_prob_old_data=0.0, solver_passes=0, phi_rtol=0.0, initial_ess=0.0, use_graph=0, stop_after_stage=0,
                    continue_run=False):
        rc = self._run_config(n_blocks, n_mh_steps, lam, n_phi, resampling_method, threshold_ratio, c, alpha, target,
                              use_fixed_schedule, tempering_target, prior_weight, log_prob_old_data, solver_passes, 0, use_graph, phi_rtol,
                              initial_ess)
        rc.stop_after_stage, rc.continue_run = int(stop_after_stage), int(bool(continue_run))
        res = _lib.Result()
        self._checked(self._L.smcmi_run_sharded(self._h, C.byref(rc), C.byref(res)))
        out = self._result(res)
        out["paused"] = bool(res.paused)
        return out

    def stages_held(self):
        """Stages the handle holds records / history columns for (smcmi_stages_held): what the getters below copy."""
        k = C.c_int32()
        check(self._L.smcmi_stages_held(self._h, C.byref(k)))
        return int(k.value)

    def stage_records(self, n_stages):
        # the library copies as many records as the handle holds (at most max_stages), whatever the caller expects
        cap = max(int(n_stages), self.stages_held(), 1)
        phi, ess, c, acc = (np.zeros(cap) for _ in range(4))
        rs = np.zeros(cap, dtype=np.int32)
        check(self._L.smcmi_get_stage_records(self._h, _d(phi), _d(ess), _d(c), _d(acc), _i(rs)))
        k = int(n_stages)
        return dict(schedule=phi[:k].copy(), ess=ess[:k].copy(), c_hist=c[:k].copy(), accept_hist=acc[:k].copy(), resampled=rs[:k].copy())

    def history(self, n_stages):
        # the library copies as many columns as the handle holds, whatever the caller expects (smcmi_stages_held)
        cap = max(int(n_stages), self.stages_held(), 1)
        w, W = np.empty((self.n, cap), order="F"), np.empty((self.n, cap), order="F")
        check(self._L.smcmi_get_history(self._h, _d(w), _d(W)))
        k = int(n_stages)
        return (w, W) if k == cap else (np.asfortranarray(w[:, :k]), np.asfortranarray(W[:, :k]))

    def sync(self):
        check(self._L.smcmi_sync(self._h))

    # ---- shard-level calls (multi-GPU hosts that drive the stage loop themselves; tests/shard_orchestrator.py) ------------
    tensor_device = "cuda"

    def _comm(self, count):
        out = np.empty(count)
        check(self._L.smcmi_comm_read(self._h, _d(out), count))
        return out

    def shard_ess_sums(self, phis, phi_prev):
        phis = _f64(np.atleast_1d(phis))
        check(self._L.smcmi_shard_ess_partial(self._h, _d(phis), phis.size, phi_prev))
        c = self._comm(2 * _lib.MAX_CAND)
        return c[:phis.size].copy(), c[_lib.MAX_CAND:_lib.MAX_CAND + phis.size].copy()

    def shard_correct(self, phi_n, phi_prev, prior_weight, log_prob_old_data, stage_col):
        check(self._L.smcmi_shard_correct_partial(self._h, phi_n, phi_prev, prior_weight, log_prob_old_data, stage_col))
        return self._comm(2)

    def shard_normalize_moments(self, sum_unnorm, resampled, shift, stage_col):
        sh = _f64(shift)
        check(self._L.smcmi_shard_normalize_moments_partial(self._h, sum_unnorm, int(resampled), _d(sh), stage_col))
        return self._comm((self.d + 1) * (self.d + 2) // 2)

    def shard_mutate(self, mu_free, Sigma_free, block_ptr, blocks_free, phi_n, phi_prev, c, alpha, n_mh_steps, stage):
        mu, S = _f64(mu_free), _f64(Sigma_free)
        bp, bf = np.ascontiguousarray(block_ptr, dtype=np.int32), np.ascontiguousarray(blocks_free, dtype=np.int32)
        check(self._L.smcmi_shard_mutate_partial(self._h, _d(mu), _d(S), _i(bp), _i(bf), bp.size - 1, phi_n, phi_prev, c, alpha,
                                                 n_mh_steps, stage))
        return float(self._comm(1)[0])

    def _dev_tensor(self, ptr, shape):
        import torch

        class _A:
            pass

        a = _A()
        a.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f8", "data": (int(ptr), False), "version": 2}
        return torch.as_tensor(a, device="cuda")

    def cloud_tensor(self):
        """Zero-copy torch view [R, n] of the current cloud buffer (rows = columns of cloud.particles)."""
        ptr, ld = C.c_void_p(), C.c_int64()
        check(self._L.smcmi_cloud_device_ptr(self._h, C.byref(ptr), C.byref(ld)))
        return self._dev_tensor(ptr.value, (self.R, self.n))

    def shard_resample(self, full_weights, full_cloud, method, stage):
        """full_weights: torch cuda tensor [N]; full_cloud: torch cuda tensor [R, N] (all-gathered)."""
        if getattr(full_cloud, "is_cuda", False):
            # the gathered tensors were produced on torch's stream; the handle's own stream is non-blocking and would not wait for it
            import torch

            torch.cuda.current_stream(full_cloud.device).synchronize()
        anc = np.empty(self.n, dtype=np.int64)
        check(self._L.smcmi_shard_resample(self._h, C.c_void_p(full_weights.data_ptr()), C.c_void_p(full_cloud.data_ptr()),
                                           _lib.RESAMPLE[method], stage, anc.ctypes.data_as(lp)))
        return anc


def torch_dist_host_comm(group=None):
    """(allgather, alltoallv, barrier) for Engine.comm_init_host on torch.distributed CPU tensors (gloo, or any backend that moves host
    tensors): what a Julia host would do with Distributed.jl, a C host with MPI."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)

    def allgather(send):
        t = torch.from_numpy(np.ascontiguousarray(send))
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t, group=group)
        return np.concatenate([o.numpy() for o in outs])

    def alltoallv(sends, recv_counts):
        # gloo has no all_to_all for CPU tensors in every build: pairwise isend / irecv in a fixed order
        recvs = [torch.empty(int(c), dtype=torch.float64) for c in recv_counts]
        reqs = []
        for p in range(world):
            if p == rank:
                continue
            if recv_counts[p]:
                reqs.append(dist.irecv(recvs[p], src=dist.get_global_rank(group, p) if group is not None else p, group=group))
            if len(sends[p]):
                reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(sends[p])), dst=dist.get_global_rank(group, p) if group is not None else p, group=group))
        for r in reqs:
            r.wait()
        return [r.numpy() for r in recvs]

    def barrier():
        dist.barrier(group=group)

    return allgather, alltoallv, barrier


def debug_proposal_densities(para_draw, para_subset, mu, Sigma, c, alpha):
    """compute_proposal_densities (src/helpers.jl:128-164) through the dense mixture code of the alpha < 1 mutation kernels
    (smcmi_debug_proposal_densities): (q0, q1) as the reference returns them."""
    pd, ps, m = _f64(para_draw), _f64(para_subset), _f64(mu)
    S = np.ascontiguousarray(np.asarray(Sigma, dtype=np.float64))
    assert S.shape == (m.size, m.size) and pd.size == m.size and ps.size == m.size
    q0, q1 = C.c_double(), C.c_double()
    check(_lib.lib().smcmi_debug_proposal_densities(_d(pd), _d(ps), _d(m), _d(S), m.size, float(c), float(alpha), C.byref(q0), C.byref(q1)))
    return q0.value, q1.value


def comm_unique_id():
    """ncclGetUniqueId through the library (call on rank 0, broadcast the 128 bytes to the other ranks)."""
    buf = C.create_string_buffer(128)
    check(_lib.lib().smcmi_comm_unique_id(buf))
    return buf.raw


def run_group(engines, n_blocks=1, n_mh_steps=1, lam=2.1, n_phi=300, resampling_method="systematic", threshold_ratio=0.5, c=0.5,
              alpha=1.0, target=0.25, use_fixed_schedule=True, tempering_target=0.97, prior_weight=0.0, log_prob_old_data=0.0,
              solver_passes=0, phi_rtol=0.0, initial_ess=0.0, stop_after_stage=0, continue_run=False):
    """Drive several shard engines of this process in lock step (smcmi_run_group)."""
    e0 = engines[0]
    rc = e0._run_config(n_blocks, n_mh_steps, lam, n_phi, resampling_method, threshold_ratio, c, alpha, target, use_fixed_schedule,
                        tempering_target, prior_weight, log_prob_old_data, solver_passes, 0, 0, phi_rtol, initial_ess)
    rc.stop_after_stage, rc.continue_run = int(stop_after_stage), int(bool(continue_run))
    res = _lib.Result()
    arr = (C.c_void_p * len(engines))(*[e._h for e in engines])
    check(_lib.lib().smcmi_run_group(arr, len(engines), C.byref(rc), C.byref(res)))
    out = Engine._result(res)
    out["paused"] = bool(res.paused)
    return out
