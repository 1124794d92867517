cd $GRAFT_REPO_ROOT; python scripts/exp_read_ceiling.py 2>&1 | tail -4
