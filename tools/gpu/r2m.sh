set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2m; mkdir -p $O
cd $R
for ss in 1 0; do
  SISR_WGRAD_SIDE_STREAM=$ss python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-timing > $O/b32_side$ss.json 2>/dev/null
  SISR_WGRAD_SIDE_STREAM=$ss python bench.py --batch 16 --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-timing --graph off > $O/b16_side$ss.json 2>/dev/null
done
python -c "
import json,glob
for f in sorted(glob.glob('$O/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f.split('/')[-1], round(d['value'],2), round(d['ms_per_step'],1))
"
