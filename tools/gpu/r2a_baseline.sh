set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2a; mkdir -p $O
cd $R
python bench.py --workload qrcan --batch 4 --steps 10 --warmup 3 --no-cpu-baseline > $O/b4_eager.json 2>$O/b4_eager.err
python bench.py --workload qrcan --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --graph > $O/b4_graph.json 2>$O/b4_graph.err
python tools/kbench.py --batch 4 --iters 50 --variants 5,6 --rounds 3 > $O/kbench_b4.jsonl 2>$O/kbench_b4.err
python tools/kbench.py --batch 32 --iters 20 > $O/kbench_b32.jsonl 2>$O/kbench_b32.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b4 -- python3 $R/bench.py --workload qrcan --batch 4 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --graph > $O/prof_b4.log 2>&1
ls -R $O/prof | head -30
