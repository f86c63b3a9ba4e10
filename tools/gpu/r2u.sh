set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2u; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_gpu.py -m gpu -q --capture=sys -k "rgb_side or conv_head_tail or rcan_reduced" > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -3 $O/t.log
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timing > $O/prof.log 2>&1
cd $R
python tools/rocpd_stats.py $O/prof/p_results.db > $O/kernel_stats.csv
grep -i "rgb_out\|cout3" $O/kernel_stats.csv
