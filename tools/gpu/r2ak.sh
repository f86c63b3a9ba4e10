set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2ak; mkdir -p $O
cd $R
for nj in 8 10 12; do
SISR_BATCH_WGRAD_JOBS=$nj python bench.py --workload qrcan --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --force-dp --no-kernel-timing > $O/b4_dp_nj$nj.json 2>/dev/null
done
SISR_BATCH_WGRAD_JOBS=4 python bench.py --workload rcan --batch 8 --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary > $O/b8_nj4.json 2>/dev/null
SISR_BATCH_WGRAD_JOBS=8 python bench.py --workload rcan --batch 8 --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-secondary > $O/b8_nj8.json 2>/dev/null
python -c "
import json,glob
for f in sorted(glob.glob('$O/b*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f.split('/')[-1], round(d['value'],2), round(d['ms_per_step'],1))
"
