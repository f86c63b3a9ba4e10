set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2ah; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_gpu.py -m gpu -q --capture=sys -k "batched_weight or fused_group or two_ranks or rcab or rcan_reduced or qrcan_reduced or trajectory" > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -2 $O/t.log
python bench.py --workload qrcan --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --force-dp --no-kernel-timing > $O/b4_dp.json 2>/dev/null
python -c "
import json
d=json.loads([l for l in open('$O/b4_dp.json') if l.startswith('{')][-1]); print('b4 dp', round(d['value'],2), round(d['ms_per_step'],1), d['config'].get('final_loss'))"
