set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2g; mkdir -p $O
cd $R
python -m pytest tests/test_degrade_gpu.py tests/test_bf16x3_gpu.py -m gpu -q --capture=sys > $O/tests.log 2>&1 || true
tail -6 $O/tests.log
K="--batch 32 --iters 20 --precision bf16x3 --only conv_v4,conv_relu_gap,conv_dgrad2,conv_res,wgrad"
python tools/kbench.py $K > $O/kb_bd4.jsonl 2>/dev/null
SISR_X3_BD=6 python tools/kbench.py $K > $O/kb_bd6.jsonl 2>/dev/null
cat $O/kb_bd4.jsonl $O/kb_bd6.jsonl
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum SQ_INST_LEVEL_LDS"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- python3 $R/tools/kbench.py --batch 32 --iters 3 --precision bf16x3 --only conv_v4,wgrad > $O/pmc_$tag.log 2>&1 || tail -3 $O/pmc_$tag.log
done
ls $O
