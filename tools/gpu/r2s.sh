set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2s; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q --capture=sys > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
