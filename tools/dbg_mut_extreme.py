"""Stage 58 of sweep trial (7, 49) in isolation: oracle cloud after 57 stages -> correction -> mutation, device vs oracle (development)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smc_jl_amd import Engine
from tests import models
from oracle import oracle as orc
n, d, seed = 6000, 9, 379
kw = {'n_blocks': 2, 'n_mh_steps': 1, 'alpha': 0.9, 'use_fixed_schedule': True, 'n_phi': 60, 'tempering_target': 0.9, 'resampling_method': 'multinomial', 'threshold_ratio': 0.8}
old_T = int(sys.argv[1]) if len(sys.argv) > 1 else 30
spec = models.linmodel_spec(T=100, old_T=old_T)
m = models.oracle_model(spec)
e = Engine(n, d, seed=seed, max_stages=100)
e.set_model(spec); e.init_from_prior()
P = np.asfortranarray(e.download_cloud()); e.close()
K = int(os.environ.get("K", "57"))
import ctypes as C
cfg = orc._RunConfig(n, kw["n_blocks"], kw["n_mh_steps"], 2.1, kw["n_phi"], orc.RESAMPLE[kw["resampling_method"]], kw["threshold_ratio"], 0.5,
                     kw["alpha"], 0.25, 1, kw["tempering_target"], 0.0, 0.0, seed, K, 8, 0.0)
sc, es, cs, ac = (np.zeros(K + 1) for _ in range(4))
rf = np.zeros(K + 1, dtype=np.int32)
res = orc._RunResult()
ms = m.struct()
rc = orc.lib().orc_smc_run(C.byref(ms), C.byref(cfg), orc._d(P), orc._d(sc), orc._d(es), orc._d(cs), orc._d(ac), orc._i(rf), None, None, C.byref(res))
print("oracle rc", rc, "c hist tail", cs[K - 3:K], "ess tail", es[K - 3:K], "acc tail", ac[K - 3:K])
print("moved so far", (P[:, d + 2] != 0).sum(), "W>0", (P[:, d + 4] > 0).sum(), "sumW", P[:, d + 4].sum(), "loglh range", P[:, d].min(), P[:, d].max())
sched = (np.arange(60) / 59.0) ** 2.1
phi, phi1 = sched[K], sched[K - 1]
c = float(os.environ.get("C", "0.03144082937802073"))
print("sched check", sc[K - 1], phi1)
Pc = orc.correct(P.copy(order="F"), phi, phi1)
Pc = Pc[0] if isinstance(Pc, tuple) else Pc
print("after correction: ESS", Pc[:, d + 4].sum() ** 2 / (Pc[:, d + 4] ** 2).sum())
mu, S = orc.weighted_mean(Pc), orc.weighted_cov(Pc)
free = np.arange(d, dtype=np.int32)
stage = K + 1
bf, ba, bp = orc.generate_blocks(d, 2, free, seed, stage)
Q = orc.mutate_cloud(m, Pc.copy(order="F"), mu, S, bf, ba, bp, phi, phi1, c, 0.9, 1, seed, stage, n_threads=8)
e = Engine(n, d, seed=seed, max_stages=10)
e.set_model(spec); e.upload_cloud(Pc)
e.mutate(mu, S, bp, bf, phi, phi1, c, 0.9, 1, stage)
D = e.download_cloud(); e.close()
acc_o, acc_d = Q[:, d + 3], D[:, d + 3]
diff = np.nonzero(acc_o != acc_d)[0]
print(" stage", stage, "c", c, "phi", phi, "accepted oracle", (acc_o > 0).sum(), "device", (acc_d > 0).sum(), "differing", diff[:10], len(diff))
for i in diff[:4]:
    print("  i", i, "before      ", Pc[i, :d], "ll", Pc[i, d], "lp", Pc[i, d + 1], "old", Pc[i, d + 2], "W", Pc[i, d + 4])
    print("  i", i, "oracle theta", Q[i, :d], "ll", Q[i, d], "old", Q[i, d + 2])
    print("  i", i, "device theta", D[i, :d], "ll", D[i, d], "old", D[i, d + 2])

# ---- the in-run device: which particles moved at stage 58, and were their proposals the oracle's?
e = Engine(n, d, seed=seed, max_stages=100)
e.set_model(spec); e.init_from_prior()
try:
    e.run(**kw)
except Exception as ex:
    print("device:", ex)
G = e.download_cloud()
mm, SS = e.moments()
e.close()
moved = np.nonzero(G[:, d + 2] != 0)[0]
print("device in-run moved", moved, "their W", G[moved, d + 4])
print("device moments (current cloud) vs oracle mean diff", np.abs(mm - mu).max())
for i in moved[:4]:
    print(" pid", i, "before", Pc[i, :d], "ll", Pc[i, d], "lp", Pc[i, d + 1])
    print(" pid", i, "device", G[i, :d], "ll", G[i, d], "lp", G[i, d + 1], "old", G[i, d + 2], "acc", G[i, d + 3])
    for t in range(2):
        idx = ba[bp[t]:bp[t + 1]]; fidx = bf[bp[t]:bp[t + 1]]
        prop = orc.mixture_draw(Pc[i, idx], mu[fidx], S[np.ix_(fidx, fidx)], c, 0.9, seed, int(i), stage, t)
        print("   block", t, "idx", idx, "oracle proposal", prop)
