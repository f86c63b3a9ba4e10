set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2aa; mkdir -p $O
cd $R
export SISR_HIP_LIB=$R/super-resolution-meta-attention-networks_amd/libsisr_hip_diag.so
for w in 512 256 384 768 1024; do
  for b in 4 32; do
    echo "wgs=$w batch=$b" >> $O/wgrad_split.log
    SISR_DIAG_WGRAD_WGS=$w python tools/kbench.py --batch $b --iters 30 --only wgrad --variants 4 >> $O/wgrad_split.log 2>&1
  done
done
cat $O/wgrad_split.log | grep -v "^$" | tail -40
