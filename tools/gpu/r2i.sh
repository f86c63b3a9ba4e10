set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2i; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q --capture=sys > $O/pytest.log 2>&1 || tail -40 $O/pytest.log
tail -3 $O/pytest.log
SISR_HIP_LIB=$R/super-resolution-meta-attention-networks_amd/libsisr_hip_diag.so python tools/x3_phases.py 32 > $O/x3_phases.log 2>&1 || tail -5 $O/x3_phases.log
cat $O/x3_phases.log | tail -6
python bench.py --steps 10 --warmup 3 > $O/default.json 2>$O/default.err || { tail -20 $O/default.err; exit 1; }
python bench.py --workload qrcan --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --force-dp > $O/b4_dp_graph.json 2>$O/b4.err || tail -5 $O/b4.err
python -c "
import json
for f in ('default','b4_dp_graph'):
    d=json.loads([l for l in open('$O/'+f+'.json') if l.startswith('{')][-1])
    print(f, round(d['value'],2), round(d['ms_per_step'],1), [(x['family'][:18], round(x['avg_launch_us'],1)) for x in d['roofline']['families']], {k:round(d[k]['value'],1) for k in ('meta_rcan','bf16x3') if k in d})
"
