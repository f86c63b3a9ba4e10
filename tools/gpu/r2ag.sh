set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2ag; mkdir -p $O
cd $R
for px in 131072 1000000; do
SISR_BATCH_WGRAD_MAX_PIXELS=$px python bench.py --workload rcan --batch 16 --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-secondary > $O/b16_px$px.json 2>/dev/null
SISR_BATCH_WGRAD_MAX_PIXELS=$px python bench.py --workload rcan --batch 32 --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-secondary > $O/b32_px$px.json 2>/dev/null
done
SISR_BATCH_WGRAD_MAX_PIXELS=1000000 SISR_WGRAD_SIDE_STREAM=0 python bench.py --workload rcan --batch 32 --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-secondary > $O/b32_px1000000_noside.json 2>/dev/null
python tools/sftmd_bench.py > $O/sftmd.json 2>/dev/null
python -c "
import json,glob
for f in sorted(glob.glob('$O/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f.split('/')[-1], round(d['value'],2), round(d['ms_per_step'],1))
"
