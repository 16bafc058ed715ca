#!/bin/bash
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S tools/exp/kaltest.hip -o /tmp/kal.s 2>&1 | grep error
python tools/isa_extract.py /tmp/kal.s _ZN5smcmiL12kalman_lgss2 /tmp/kal_f.s | cut -c1-400
grep -E "num_vgpr|num_agpr|scratch|private_seg" /tmp/kal.s | grep -i "kalman" | head -5
python - <<'PY'
import re,collections
lines=open('/tmp/kal_f.s').read().split('\n')
labels={l.split(':')[0]:i for i,l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:',l)}
best=None
for i,l in enumerate(lines):
    m=re.search(r's_cbranch\w*\s+(\.LBB\d+_\d+)',l) or re.search(r's_branch\s+(\.LBB\d+_\d+)',l)
    if m and m.group(1) in labels and labels[m.group(1)]<i:
        sp=(labels[m.group(1)],i)
        if best is None or sp[1]-sp[0]>best[1]-best[0]: best=sp
body=[l.strip() for l in lines[best[0]:best[1]+1] if l.strip() and not l.strip().startswith(('.',';')) and not l.strip().endswith(':')]
c=collections.Counter(b.split()[0] for b in body)
print('loop instrs',sum(c.values()),'f64',sum(v for k,v in c.items() if 'f64' in k),'readlane',c['v_readlane_b32'],'accvgpr',c['v_accvgpr_read_b32']+c['v_accvgpr_write_b32'],'scratch',sum(v for k,v in c.items() if k.startswith('scratch')), 's_load', sum(v for k,v in c.items() if k.startswith('s_load')))
PY
