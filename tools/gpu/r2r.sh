set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2r; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_san_gpu.py -m gpu -q --capture=sys -x > $O/san.log 2>&1 || { tail -60 $O/san.log; exit 1; }
tail -3 $O/san.log
