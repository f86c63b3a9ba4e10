set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2y; mkdir -p $O
cd $R
SISR_PRECISION=bf16x3 timeout -k 10 1000 python -m pytest tests -m gpu -q --capture=sys > $O/fp32_suite_under_bf16x3.log 2>&1 || { tail -60 $O/fp32_suite_under_bf16x3.log; exit 1; }
tail -2 $O/fp32_suite_under_bf16x3.log
