set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2al; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_gpu.py -m gpu -q --capture=sys -x -k "gate_heads or batched_weight or fused_group or rcan_reduced or qrcan_reduced or two_ranks or graph or trajectory" > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -2 $O/t.log
for fs in 1 0; do
SISR_FUSE_SLAB_SUM=$fs python bench.py --workload qrcan --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --force-dp --no-kernel-timing > $O/b4_dp_fs$fs.json 2>/dev/null
SISR_FUSE_SLAB_SUM=$fs python bench.py --workload rcan --batch 8 --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary > $O/b8_fs$fs.json 2>/dev/null
done
python -c "
import json,glob
for f in sorted(glob.glob('$O/b*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f.split('/')[-1], round(d['value'],2), round(d['ms_per_step'],1), d['config'].get('final_loss'))
"
