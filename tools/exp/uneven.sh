#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
for N in 100000 100001 50002 100004 65538 32770 126972; do
python bench.py --nparts $N --steps 5 --warmup 1 --no-cpu 2>/dev/null | grep '^{' | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('N=$N', '%.4g'%d['value'], '%.3f'%d['ms_per_step'], d['n_stages'], d['resamples'], '%.10f'%d['logmdd_gpu'], d['roofline']['kernel'][:28], '%.1f us/stage' % (1e3*d['ms_per_step']/(d['n_stages']-1)))"
done
timeout 600 python tools/stress_segments.py 4 50002 100004 32770 2>&1 | tail -9
