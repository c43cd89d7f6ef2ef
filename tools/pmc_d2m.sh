cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_d2m_r2; rm -rf $O; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
for S in 128 256; do
S=$S timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $O/a$S -o d2m -- python tools/prof_d2m.py > $O/a$S.log 2>&1
S=$S timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d $O/b$S -o d2m -- python tools/prof_d2m.py > $O/b$S.log 2>&1
S=$S timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES --output-format csv -d $O/c$S -o d2m -- python tools/prof_d2m.py > $O/c$S.log 2>&1
done
find $O -name "*counter_collection.csv" | head; du -sh $O
