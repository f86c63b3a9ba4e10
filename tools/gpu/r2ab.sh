set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2ab; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_sftmd_gpu.py -m gpu -q --capture=sys > $O/t.log 2>&1 || { tail -70 $O/t.log; exit 1; }
tail -3 $O/t.log
