import sys,json
bad=0
for l in sys.stdin:
    l=l.strip()
    if l.startswith("{"):
        j=json.loads(l)
        flag = (not j["ok"]) or j["logmdd_err"]>1e-7
        bad += flag
        if flag or j["trial"]%6==0 or "error" in j: print(j["trial"], j["kind"], "d",j["d"],"n",j["n"],"nb",j["n_blocks"],"mh",j["n_mh_steps"],"a",j["alpha"],"fix",j["use_fixed_schedule"],j["resampling_method"][:4],"thr",j["threshold_ratio"],"ok",j["ok"],"err %.1e ess %.1e"%(j["logmdd_err"],j["ess_relerr"]),"st",j["stages"],j["rs"],j["stalls"],j.get("ms"), j.get("error",""), "BAD" if flag else "")
    else: print(l)
print("bad", bad)
