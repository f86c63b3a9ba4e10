set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2l; mkdir -p $O
cd $R
for b in 4 8 32; do python tools/pair_probe.py $b 100 >> $O/pair_probe.jsonl 2>/dev/null; done
cat $O/pair_probe.jsonl
python -m pytest tests/test_hip_gpu.py tests/test_san_gpu.py -m gpu -q --capture=sys -k "han or san or HAN or SAN" > $O/han_san.log 2>&1 || tail -20 $O/han_san.log
tail -3 $O/han_san.log
SISR_PRECISION=bf16x3 python -m pytest tests/test_hip_gpu.py tests/test_san_gpu.py tests/test_srmd_gpu.py -m gpu -q --capture=sys > $O/fp32_suite_under_x3.log 2>&1 || true
tail -3 $O/fp32_suite_under_x3.log
