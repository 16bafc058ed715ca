cd $GRAFT_REPO_ROOT
echo "=== full GPU suite"
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8
echo "=== smoke"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "=== bench default"
python bench.py 2>/dev/null | tail -1 | cut -c1-900
echo "=== shard rank gaps"
bash tools/shard_rank_prof.sh big10 250000 2>&1 | grep -v "rocclr\|k_noop\|selftest\|k_init_prior\|k2_scan\|k2_pass\|k2_reduce\|k2_gather"
