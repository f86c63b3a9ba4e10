#!/usr/bin/env bash
# One parameterised lease script for the GPU box:  gpurun -- 'bash tools/gpu/run.sh RECIPE [ARGS] [-- RECIPE [ARGS]]...'
# Outputs go to gpurun_out/<recipe>/ (merged back by gpurun); summaries worth keeping are copied to profiles/ by hand.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
DIAG=$R/super-resolution-meta-attention-networks_amd/libsisr_hip_diag.so

prof() {  # prof OUTDIR NAME python-args...   (rocprofv3 kernel-trace summary -> OUTDIR/kernel_stats_NAME.csv)
  local O=$1 name=$2; shift 2
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$O/prof_$name" -o p -- python "$@" > "$O/prof_$name.log" 2>&1 )
  python tools/rocpd_stats.py "$O/prof_$name/p_results.db" > "$O/kernel_stats_$name.csv"
  rm -rf "$O/prof_$name"
}

recipe() {
  local RECIPE=$1; shift
  local O=$R/gpurun_out/$RECIPE; mkdir -p "$O"
  case "$RECIPE" in
    tests)      # the driver's GPU tier
      timeout -k 10 1100 python -m pytest tests -m gpu -q -x --capture=sys "$@" > "$O/pytest.log" 2>&1 || { tail -60 "$O/pytest.log"; return 1; }
      tail -3 "$O/pytest.log"
      python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$O/smoke.log" 2>&1 || { tail -20 "$O/smoke.log"; return 1; }
      tail -1 "$O/smoke.log" ;;
    pytest)     # selected tests: pytest ARGS
      timeout -k 10 1100 python -m pytest -q -x --capture=sys "$@" > "$O/pytest.log" 2>&1 || { tail -80 "$O/pytest.log"; return 1; }
      tail -5 "$O/pytest.log" ;;
    timeline)   # in-kernel stamps of the fp32 conv (diagnostic library): timeline [BATCHES] [FORMS]
      for b in ${1:-4 32}; do for f in ${2:-plain mask gate}; do
        SISR_HIP_LIB=$DIAG timeout -k 10 120 python tools/conv_timeline.py $b $f > "$O/timeline_b${b}_$f.json" 2> "$O/timeline_b${b}_$f.err" || tail -5 "$O/timeline_b${b}_$f.err"
        head -1 "$O/timeline_b${b}_$f.json" | cut -c1-1400
      done; done ;;
    bf16s)      # in-kernel cycle sums of the bf16-map persistent conv (diagnostic library): bf16s [BATCHES]
      for b in ${1:-32 16}; do
        SISR_HIP_LIB=$DIAG timeout -k 10 120 python tools/bf16s_timeline.py $b > "$O/bf16s_b$b.json" 2> "$O/bf16s_b$b.err" || tail -5 "$O/bf16s_b$b.err"
        cat "$O/bf16s_b$b.json"
        SISR_HIP_LIB=$DIAG timeout -k 10 120 python tools/bf16s_timeline.py $b --wgrad > "$O/bf16s_wgrad_b$b.json" 2> "$O/bf16s_wgrad_b$b.err" || tail -5 "$O/bf16s_wgrad_b$b.err"
        cat "$O/bf16s_wgrad_b$b.json"
      done ;;
    kbench)     # kbench [BATCHES] [extra kbench args]
      local bs=${1:-4 32}; shift || true
      for b in $bs; do timeout -k 10 200 python tools/kbench.py --batch $b --iters 30 --only conv,conv_dgrad2,conv_res,conv_relu_gap,wgrad,wgrad_affine "$@" > "$O/kbench_b$b.jsonl"; cat "$O/kbench_b$b.jsonl"; done ;;
    ab)         # ab [BATCHES] [ONLY]: kbench with the library of the previous commit (_ab/libsisr_hip_old.so) and the current one, interleaved
      local bs=${1:-4 32} only=${2:-conv_relu_gap,conv_dgrad2,conv_res,wgrad}
      local OLD=$R/super-resolution-meta-attention-networks_amd/_ab/libsisr_hip_old.so
      for b in $bs; do for rep in 1 2; do for lib in old new; do
        if [ $lib = old ]; then export SISR_HIP_LIB=$OLD; else unset SISR_HIP_LIB; fi
        timeout -k 10 200 python tools/kbench.py --batch $b --iters 40 --warm 60 --only $only 2>/dev/null | sed "s/^{/{\"lib\": \"$lib\", /" >> "$O/ab_b$b.jsonl"
      done; done; done
      unset SISR_HIP_LIB
      python - "$O" $bs <<'PY'
import json, sys, collections
O = sys.argv[1]
for b in sys.argv[2:]:
    acc = collections.defaultdict(list)
    for l in open(f"{O}/ab_b{b}.jsonl"):
        if l.startswith('{'):
            d = json.loads(l); acc[(d['kernel'], d['lib'])].append(d['us'])
    for k in sorted({k for k, _ in acc}):
        o, n = min(acc[(k, 'old')]), min(acc[(k, 'new')])
        print(f"B={b} {k:16s} old {o:8.1f} us  new {n:8.1f} us  ({100 * (o / n - 1):+.1f} %)")
PY
      ;;
    peak)       # sustained fp32 MFMA rate and in-kernel clock, trivial and random operands
      SISR_HIP_LIB=$DIAG timeout -k 10 200 python tools/mfma_peak.py > "$O/mfma_peak.jsonl" 2>&1; cat "$O/mfma_peak.jsonl" ;;
    fill)       # cost of filler instructions beside the fp32 MFMA stream
      SISR_HIP_LIB=$DIAG timeout -k 10 300 python tools/mfma_fill.py > "$O/mfma_fill.jsonl" 2>&1; cat "$O/mfma_fill.jsonl" ;;
    copies)
      timeout -k 10 300 python tools/copy_probe.py 4 qrcan dp > "$O/copies_qrcan_b4.txt" 2>&1 || true
      tail -70 "$O/copies_qrcan_b4.txt" ;;
    bench)      # bench TAG bench.py-args...  -> gpurun_out/bench/bench_TAG.json
      local tag=$1; shift
      timeout -k 10 900 python bench.py "$@" > "$O/bench_$tag.json" 2> "$O/bench_$tag.err" || { tail -20 "$O/bench_$tag.err"; return 1; }
      python - "$O/bench_$tag.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1].split('/')[-1], 'value', round(d['value'], 2), 'ms', round(d['ms_per_step'], 2), 'frac', (d.get('roofline') or {}).get('frac'))
for f in (d.get('roofline') or {}).get('families') or []:
    print('   ', f['family'][:50], f['launches_per_step'], round(f['avg_launch_us'], 1), round(f['frac'], 3))
for k in ('config4_point', 'han_bf16', 'meta_rcan', 'bf16x3'):
    if k in d:
        print('   ', k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in d[k].items() if not isinstance(b, (dict, list, str))})
PY
      ;;
    profile)    # profile TAG bench.py-args...  -> rocprofv3 kernel summary of that bench command
      local tag=$1; shift
      prof "$O" "$tag" "$R/bench.py" "$@"
      head -14 "$O/kernel_stats_$tag.csv" | cut -c1-170 ;;
    anatomy)    # anatomy TAG script args...: kernel trace of `python script args` -> where the last step's wall time goes
      local tag=$1; shift
      ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d "$O/prof_$tag" -o p -- python "$R/$1" "${@:2}" > "$O/prof_$tag.log" 2>&1 )
      python tools/rocpd_step.py "$O/prof_$tag/p_results.db" > "$O/anatomy_$tag.txt"
      python tools/rocpd_gaps.py "$O/prof_$tag/p_results.db" 15 >> "$O/anatomy_$tag.txt"
      python tools/rocpd_window.py "$O/prof_$tag/p_results.db" 300 50 > "$O/window_$tag.txt"; python tools/rocpd_window.py "$O/prof_$tag/p_results.db" 1200 50 >> "$O/window_$tag.txt"
      python tools/rocpd_stats.py "$O/prof_$tag/p_results.db" > "$O/kernel_stats_$tag.csv"
      rm -rf "$O/prof_$tag"; cat "$O/anatomy_$tag.txt" | cut -c1-160 ;;
    lanes)      # lanes [BATCHES] [extra lane_probe args]: half-batch conv chains on parallel graph branches
      local bs=${1:-4 8}; shift || true
      for b in $bs; do timeout -k 10 300 python tools/lane_probe.py --batch $b "$@" > "$O/lanes_b$b.jsonl" 2> "$O/lanes_b$b.err" || tail -5 "$O/lanes_b$b.err"; cat "$O/lanes_b$b.jsonl"; done ;;
    pmc)        # pmc TAG COUNTER kbench-args...: one rocprofv3 --pmc pass over tools/kbench.py -> per-kernel mean of the counter
      local tag=$1 ctr=$2; shift 2
      ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$O/pmc_${tag}_$ctr" -o p -- python "$R/tools/kbench.py" "$@" > "$O/pmc_${tag}_$ctr.log" 2>&1 )
      python - "$O/pmc_${tag}_$ctr" $ctr <<'PY' > "$O/pmc_${tag}_$ctr.csv"
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == sys.argv[2]:
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
print("kernel,launches,mean_" + sys.argv[2])
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print('"%s",%d,%.1f' % (k[:120], len(v), sum(v) / len(v)))
PY
      rm -rf "$O/pmc_${tag}_$ctr"; head -6 "$O/pmc_${tag}_$ctr.csv" | cut -c1-200 ;;
    pmcb)       # pmcb TAG COUNTER bench-args...: one rocprofv3 --pmc pass over bench.py -> per-kernel mean of the counter
      local tag=$1 ctr=$2; shift 2
      ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$O/pmc_${tag}_$ctr" -o p -- python "$R/bench.py" "$@" > "$O/pmc_${tag}_$ctr.log" 2>&1 )
      python - "$O/pmc_${tag}_$ctr" $ctr <<'PY' > "$O/pmc_${tag}_$ctr.csv"
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == sys.argv[2]:
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
print("kernel,launches,mean_" + sys.argv[2])
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print('"%s",%d,%.1f' % (k[:140], len(v), sum(v) / len(v)))
PY
      rm -rf "$O/pmc_${tag}_$ctr"; head -8 "$O/pmc_${tag}_$ctr.csv" | cut -c1-200 ;;
    sh)         # sh 'command'   (ad-hoc)
      bash -c "$1" > "$O/sh.log" 2>&1 || { tail -40 "$O/sh.log"; return 1; }
      tail -40 "$O/sh.log" ;;
    *) echo "unknown recipe $RECIPE"; return 2 ;;
  esac
}

# several recipes in one lease: separated by `--`
args=()
for a in "$@"; do
  if [ "$a" = "--" ]; then recipe "${args[@]}"; args=(); else args+=("$a"); fi
done
[ ${#args[@]} -gt 0 ] && recipe "${args[@]}"
