set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q --capture=sys > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for fork in 1 0; do
SISR_GRAPH_FORK=$fork python bench.py --workload qrcan --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --force-dp > $O/b4_dp_graph_fork$fork.json 2>$O/b4_fork$fork.err || { tail -20 $O/b4_fork$fork.err; exit 1; }
done
python bench.py --workload qrcan --batch 4 --steps 10 --warmup 3 --no-cpu-baseline --graph off > $O/b4_eager.json 2>$O/b4_eager.err
python bench.py --steps 10 --warmup 3 > $O/default.json 2>$O/default.err || { tail -20 $O/default.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b4 -- python3 $R/bench.py --workload qrcan --batch 4 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --force-dp > $O/prof_b4.log 2>&1
