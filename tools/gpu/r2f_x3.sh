set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2f; mkdir -p $O
cd $R
python -m pytest tests/test_bf16x3_gpu.py -m gpu -q --capture=sys -s > $O/x3_tests.log 2>&1 || true
tail -25 $O/x3_tests.log
python tools/kbench.py --batch 32 --iters 20 --precision bf16x3 --only conv_v4,conv_dgrad2_v4,conv_relu_gap,conv_dgrad2,conv_res,wgrad,wgrad_affine > $O/kbench_x3_b32.jsonl 2>$O/kbench_x3.err || tail -5 $O/kbench_x3.err
cat $O/kbench_x3_b32.jsonl
python bench.py --precision bf16x3 --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_x3_rcan_b32.json 2>$O/bench_x3.err || tail -5 $O/bench_x3.err
SISR_X3_WGRAD=0 python bench.py --precision bf16x3 --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_x3_fp32wgrad_rcan_b32.json 2>>$O/bench_x3.err || true
python bench.py --precision bf16x3 --workload qrcan --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_x3_qrcan_b32.json 2>>$O/bench_x3.err || true
SISR_PRECISION=bf16x3 python -m pytest tests/test_hip_gpu.py tests/test_san_gpu.py tests/test_srmd_gpu.py -m gpu -q --capture=sys > $O/fp32_suite_under_x3.log 2>&1 || true
tail -15 $O/fp32_suite_under_x3.log
