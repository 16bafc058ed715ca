"""Development: the lane-split Kalman mutation (four lanes per particle) against one thread per particle - same run, values and time.
    python tools/kalman_lanes.py [n_parts]      (runs itself twice with SMCMI_KALMAN_LANES = 4 / 1)"""
import hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(n):
    import numpy as np
    from smc_jl_amd import Engine
    from tests import models
    sp = models.kalman_spec(T=80, old_T=40)
    e = Engine(n, 13, seed=17, max_stages=400, store_history=False)
    e.set_model(sp); e.init_from_prior()
    P0 = e.download_cloud()
    # one mutation stage by hand: fixed 1-stage schedule
    kw = dict(n_phi=12, use_fixed_schedule=True, n_blocks=1, n_mh_steps=1, alpha=0.9)
    e.run(**kw)                                    # warm-up
    e.upload_cloud(P0)
    t0 = time.perf_counter(); r = e.run(**kw); dt = time.perf_counter() - t0
    P = e.download_cloud()
    np.save(os.path.join(ROOT, "gpurun_out", "kl_%s_%d.npy" % (os.environ.get("SMCMI_KALMAN_LANES", "4"), n)), P[:4096])
    print("RESULT " + json.dumps(dict(n=n, lanes=os.environ.get("SMCMI_KALMAN_LANES", "4"), n_stages=r["n_stages"], logmdd=r["logmdd"], ms=dt * 1e3,
                                      kernel_ms=r.get("seconds", None), acc=float(P[:, 16].mean()), sha=hashlib.sha256(P.tobytes()).hexdigest()[:12])))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "worker":
        worker(int(sys.argv[1]))
    else:
        import numpy as np
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        for n in [int(a) for a in sys.argv[1:]] or [12500]:
            for lanes in ("4", "1"):
                p = subprocess.run([sys.executable, __file__, str(n), "worker"], env=dict(os.environ, SMCMI_KALMAN_LANES=lanes), capture_output=True, text=True, timeout=900)
                print(p.stdout.strip()[-600:], p.stderr.strip()[-1500:])
            a = np.load(os.path.join(ROOT, "gpurun_out", "kl_4_%d.npy" % n)); b = np.load(os.path.join(ROOT, "gpurun_out", "kl_1_%d.npy" % n))
            same = np.all(a[:, :13] == b[:, :13], axis=1)
            print("n", n, "same theta rows", same.mean(), "max rel loglh diff on same rows", np.max(np.abs(a[same, 13] - b[same, 13]) / (1 + np.abs(b[same, 13]))),
                  "old", np.max(np.abs(a[same, 15] - b[same, 15]) / (1 + np.abs(b[same, 15]))))
