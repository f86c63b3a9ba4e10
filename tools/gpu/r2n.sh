set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2n; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_gpu.py -m gpu -q --capture=sys -x -k "tails or fused_group or g1_rcab or rcan_reduced or qrcan_reduced" > $O/tails.log 2>&1 || { tail -40 $O/tails.log; exit 1; }
tail -3 $O/tails.log
timeout -k 10 900 python -m pytest tests -m gpu -q --capture=sys > $O/pytest.log 2>&1 || tail -40 $O/pytest.log
tail -3 $O/pytest.log
for t in auto 0 1; do
SISR_CA_TAIL=$t python bench.py --workload qrcan --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --force-dp --no-kernel-timing > $O/b4_tail_$t.json 2>/dev/null
SISR_CA_TAIL=$t python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-timing > $O/b32_tail_$t.json 2>/dev/null
done
python -c "
import json,glob
for f in sorted(glob.glob('$O/b*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f.split('/')[-1], round(d['value'],2), round(d['ms_per_step'],1))
"
