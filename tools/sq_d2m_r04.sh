cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04i; rm -rf $O; mkdir -p $O
SQA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU"
SQB="SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"
for S in 128 256; do
  S=$S REPS=6 timeout 300 rocprofv3 --kernel-trace --pmc $SQA --output-format csv -d $O/sqa_s$S -o d2m -- python tools/prof_d2m.py > $O/loga_s$S.txt 2>&1
  S=$S REPS=6 timeout 300 rocprofv3 --kernel-trace --pmc $SQB --output-format csv -d $O/sqb_s$S -o d2m -- python tools/prof_d2m.py > $O/logb_s$S.txt 2>&1
done
python tools/summarize_sq.py $O d2m_ 2>&1 | tail -80
find $O -name "*.csv" -size +2M -delete
