set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2ah; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_gpu.py -m gpu -q --capture=sys -k "batched_weight or fused_group or two_ranks" > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -2 $O/t.log
