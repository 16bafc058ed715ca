#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for N in 140000 180000 262144 400000; do for e in 1 2; do
echo -n "N=$N engine=$e  "; SMCMI_ENGINE=$e python bench.py --steps 3 --warmup 1 --no-cpu --no-history --nparts $N 2>/dev/null | grep -o "ms_per_step[^,]*"
done; done
