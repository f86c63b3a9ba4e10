set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2o; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_sftmd_gpu.py -m gpu -q --capture=sys > $O/sftmd.log 2>&1 || { tail -60 $O/sftmd.log; }
tail -3 $O/sftmd.log
timeout -k 10 600 python -m pytest tests/test_srmd_gpu.py tests/test_hip_gpu.py -m gpu -q --capture=sys -x > $O/regress.log 2>&1 || { tail -40 $O/regress.log; exit 1; }
tail -3 $O/regress.log
