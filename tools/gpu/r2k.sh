set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2k; mkdir -p $O
cd $R
python -m pytest tests/test_bf16x3_gpu.py -m gpu -q --capture=sys -s > $O/x3_tests.log 2>&1 || true
tail -4 $O/x3_tests.log
SISR_PRECISION=bf16x3 python -m pytest tests/test_hip_gpu.py tests/test_san_gpu.py tests/test_srmd_gpu.py -m gpu -q --capture=sys > $O/fp32_suite_under_x3.log 2>&1 || true
tail -4 $O/fp32_suite_under_x3.log
python tools/kbench.py --batch 32 --iters 20 --precision bf16x3 --only conv_v4,conv_relu_gap,conv_dgrad2,conv_res,wgrad > $O/kb_x3.jsonl 2>/dev/null
cat $O/kb_x3.jsonl
SISR_HIP_LIB=$R/super-resolution-meta-attention-networks_amd/libsisr_hip_diag.so python tools/x3_phases.py 32 > $O/x3_phases.log 2>&1 || tail -5 $O/x3_phases.log
tail -5 $O/x3_phases.log
python bench.py --precision bf16x3 --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_x3_rcan_b32.json 2>$O/bench_x3.err || tail -5 $O/bench_x3.err
python -c "
import json
d=json.loads([l for l in open('$O/bench_x3_rcan_b32.json') if l.startswith('{')][-1]); print('x3 rcan', round(d['value'],1), round(d['ms_per_step'],1))
"
