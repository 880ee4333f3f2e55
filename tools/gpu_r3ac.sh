R=$GRAFT_REPO_ROOT; cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
