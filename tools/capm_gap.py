"""Config 4 (CAPM, fixed schedule, 3 MH steps) at full size against the oracle, stage by stage.

One subprocess per (library build, engine setting, N) - the environment is read once per process.  Each worker
  1. draws the initial cloud on the device, hands the SAME cloud to the oracle, runs both whole loops and compares the per-stage
     records (ESS, acceptance, resample flags) and the per-stage log-MDD increments (from the w / W history);
  2. brackets single stages of the device run with stop_after_stage / continue_run and lets the oracle repeat exactly that stage on the
     downloaded cloud: flipped MH decisions and differing rows of ONE stage, free of whatever accumulated before it.
usage: python tools/capm_gap.py [N ...]   (default 200000); CAPM_GAP_CASES=strict1,prod1,strictdef,proddef selects the settings"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STRICT = os.path.join(ROOT, "smc.jl_amd", "csrc", "libsmcmi_strict.so")

W = r'''
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
from oracle import oracle as orc
from smc_jl_amd import Engine
from tests import models
n, seed, nbr = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
spec = models.capm_spec(); m = models.oracle_model(spec); d = 9
kw = dict(use_fixed_schedule=True, n_phi=300, lam=2.1, n_mh_steps=3)
e = Engine(n, d, seed=seed, max_stages=300, store_history=True)
e.set_model(spec); e.init_from_prior()
P0 = e.download_cloud()
init_diff = float(np.max(np.abs(P0 - orc.initial_draw(m, n, seed=seed)) / (1 + np.abs(P0))))
g = e.run(**kw)
rec = e.stage_records(g["n_stages"])
w, Wn = e.history(g["n_stages"])
inc_g = np.log(np.sum(w[:, 1:] * Wn[:, :-1], axis=0) / n)
Pg = e.download_cloud()
r = orc.smc_run(m, P0, seed=seed, n_threads=64, history=True, **kw)
inc_c = np.log(np.sum(r["w"][:, 1:] * r["W"][:, :-1], axis=0) / n)
out = dict(n=n, seed=seed, init_diff=init_diff, logmdd_gpu=g["logmdd"], logmdd_cpu=r["logmdd"], res_gpu=g["resamples"], res_cpu=r["resamples"],
           ess_gpu=[float(x) for x in rec["ess"]], ess_cpu=[float(x) for x in r["ess"]], acc_gpu=[float(x) for x in rec["accept_hist"]],
           acc_cpu=[float(x) for x in r["accept_hist"]], resampled_gpu=[int(x) for x in rec["resampled"]], resampled_cpu=[int(x) for x in r["resampled"]],
           inc_gpu=[float(x) for x in inc_g], inc_cpu=[float(x) for x in inc_c])
# how many distinct ancestors survive: rows of the final clouds that agree
same_rows = int(np.count_nonzero(np.all(np.abs(Pg[:, :d] - r["particles"][:, :d]) <= 1e-9 * (1 + np.abs(Pg[:, :d])), axis=1)))
out["final_rows_equal"] = same_rows
# per-stage distinct-particle count of the oracle's history is not kept; the weights tell the degeneracy: max normalised weight
out["maxW_cpu"] = [float(r["W"][:, k].max()) for k in range(r["n_stages"])]
# ---- bracketed stages
e2 = Engine(n, d, seed=seed, max_stages=300, store_history=False)
e2.set_model(spec); e2.init_from_prior()
br = []
cont = False
for k in range(2, 2 + nbr):
    if not cont:
        rr = e2.run(stop_after_stage=k - 1, continue_run=False, **kw); cont = True
    A = e2.download_cloud()
    rr = e2.run(stop_after_stage=k, continue_run=True, **kw)
    B = e2.download_cloud()
    rc2 = e2.stage_records(rr["n_stages"])
    phi1, phi0, resampled, c = rc2["schedule"][k - 1], rc2["schedule"][k - 2], int(rc2["resampled"][k - 1]), rc2["c_hist"][k - 1]
    Pc, incw, nw, ess, su = orc.correct(A, phi1, phi0)
    res_cpu = int(ess < 0.5 * n)
    if res_cpu:
        idx = orc.resample(Pc[:, d + 4] / n, "systematic", seed=seed, stage=k)
        Pc = np.asfortranarray(Pc[idx]); Pc[:, d + 4] = 1.0
    mean, cov = orc.weighted_mean(Pc), orc.weighted_cov(Pc)
    S = (cov + cov.T) / 2
    bf, ba, bp = orc.generate_blocks(d, 1, m.free_inds, seed, k)
    want = orc.mutate_cloud(m, Pc, mean, S, bf, ba, bp, phi1, phi0, c, 1.0, 3, seed, k, n_threads=64)
    flips = int(np.count_nonzero(B[:, d + 3] != want[:, d + 3]))
    rows = int(np.count_nonzero(np.any(np.abs(B[:, :d + 3] - want[:, :d + 3]) > 1e-9 * (1 + np.abs(want[:, :d + 3])), axis=1)))
    wrel = float(np.max(np.abs(B[:, d + 4] - want[:, d + 4]) / (1e-300 + np.abs(want[:, d + 4]))))
    br.append(dict(stage=k, resampled_gpu=resampled, resampled_cpu=res_cpu, ess_cpu=float(ess), ess_gpu=float(rc2["ess"][k - 1]), flips=flips, rows_differ=rows, w_rel=wrel,
                   distinct=int(np.unique(B[:, 0]).size)))
out["bracket"] = br
print("RESULT " + json.dumps(out))
''' % dict(root=ROOT)


def run(env, lib, n, seed, nbr):
    envv = dict(os.environ, **env)
    if lib:
        envv["SMCMI_LIBRARY"] = lib
    p = subprocess.run([sys.executable, "-c", W, str(n), str(seed), str(nbr)], env=envv, capture_output=True, text=True, cwd=ROOT)
    if p.returncode:
        print(p.stderr[-3000:])
        raise SystemExit(1)
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])


def report(tag, o):
    import numpy as np
    eg, ec = np.array(o["ess_gpu"]), np.array(o["ess_cpu"])
    ig, ic = np.array(o["inc_gpu"]), np.array(o["inc_cpu"])
    print("== %s N=%d seed=%d: logmdd gpu %.9f cpu %.9f |diff| %.3e resamples %d/%d init_diff %.1e final rows equal %d" % (
        tag, o["n"], o["seed"], o["logmdd_gpu"], o["logmdd_cpu"], abs(o["logmdd_gpu"] - o["logmdd_cpu"]), o["res_gpu"], o["res_cpu"], o["init_diff"], o["final_rows_equal"]))
    rel = np.abs(eg - ec) / np.maximum(np.abs(ec), 1e-300)
    bad = np.nonzero(rel > 1e-9)[0]
    print("   first record with |dESS|/ESS > 1e-9:", (int(bad[0]) + 1) if bad.size else None, " min ESS cpu %.3f at stage %d" % (ec[1:].min(), int(ec[1:].argmin()) + 2))
    for k in range(min(8, len(eg))):
        print("   stage %3d ess gpu %.9g cpu %.9g acc %.6f %.6f res %d %d maxW %.4g inc %.9g %.9g" % (
            k + 1, eg[k], ec[k], o["acc_gpu"][k], o["acc_cpu"][k], o["resampled_gpu"][k], o["resampled_cpu"][k], o["maxW_cpu"][k],
            ig[k - 1] if k else 0.0, ic[k - 1] if k else 0.0))
    dinc = ig - ic
    order = np.argsort(-np.abs(dinc))[:6]
    print("   largest per-stage log-MDD increment differences (stage: gpu - cpu):", ", ".join("%d: %.3e" % (int(j) + 2, dinc[j]) for j in order), " cumulative %.4e" % dinc.sum())
    for b in o["bracket"]:
        print("   bracket stage %3d: res %d/%d ess %.6g/%.6g flips %d rows_differ %d w_rel %.2e distinct first-parameter values %d" % (
            b["stage"], b["resampled_gpu"], b["resampled_cpu"], b["ess_gpu"], b["ess_cpu"], b["flips"], b["rows_differ"], b["w_rel"], b["distinct"]))


if __name__ == "__main__":
    ns = [int(x) for x in sys.argv[1:]] or [200000]
    cases = {"strict1": ({"SMCMI_ENGINE": "1"}, STRICT), "prod1": ({"SMCMI_ENGINE": "1"}, None), "strictdef": ({}, STRICT), "proddef": ({}, None)}
    sel = os.environ.get("CAPM_GAP_CASES", "strict1,prod1").split(",")
    seed = int(os.environ.get("CAPM_GAP_SEED", "1"))
    nbr = int(os.environ.get("CAPM_GAP_BRACKET", "10"))
    allout = {}
    for n in ns:
        for c in sel:
            o = run(cases[c][0], cases[c][1], n, seed, nbr)
            report(c, o)
            allout["%s_n%d" % (c, n)] = o
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "capm_gap.json"), "w") as f:
        json.dump(allout, f)
