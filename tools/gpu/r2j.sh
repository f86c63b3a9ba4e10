set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2j; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q --capture=sys > $O/pytest.log 2>&1 || tail -40 $O/pytest.log
tail -3 $O/pytest.log
python bench.py --steps 10 --warmup 3 > $O/default.json 2>$O/default.err || { tail -20 $O/default.err; exit 1; }
python bench.py --workload qrcan --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --force-dp > $O/b4_dp_graph.json 2>$O/b4.err || tail -5 $O/b4.err
SISR_DIST_BACKEND=gloo SISR_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29911 bench.py --gpus 2 --steps 5 --warmup 2 > $O/n2_shared_gpu.json 2>$O/n2.err || tail -8 $O/n2.err
python -c "
import json
for f in ('default','b4_dp_graph','n2_shared_gpu'):
    try:
        d=json.loads([l for l in open('$O/'+f+'.json') if l.startswith('{')][-1])
    except Exception as e:
        print(f, 'no line', e); continue
    print(f, d['n_gpus'], d['scaling'], d['config']['workload'][:12], d['config']['per_gpu_batch'], round(d['value'],2), round(d['ms_per_step'],1), [(x['family'][:14], round(x['avg_launch_us'],1)) for x in d['roofline'].get('families',[])], {k:round(d[k]['value'],1) for k in ('meta_rcan','bf16x3','weak_scaling') if k in d})
"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o def -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-kernel-timing > $O/prof_def.log 2>&1 || tail -3 $O/prof_def.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python3 $R/tools/kbench.py --batch 32 --iters 3 --only conv_v4,conv_dgrad2,conv_res,wgrad > $O/pmc_$c.log 2>&1 || tail -3 $O/pmc_$c.log
done
ls $O $O/prof | head -30
