cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/sq_mesh; rm -rf $O; mkdir -p $O
SQA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU"
SQB="SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR"
for V in lat tile; do
  if [ $V = tile ]; then export SHR_MESH_LATTICE=0; else unset SHR_MESH_LATTICE; fi
  timeout 200 rocprofv3 --kernel-trace --pmc $SQA --output-format csv -d $O/a_$V -o m -- python tools/prof_mesh.py > $O/a_$V.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc $SQB --output-format csv -d $O/b_$V -o m -- python tools/prof_mesh.py > $O/b_$V.log 2>&1
done
python tools/summarize_sq.py $O mesh_ > $O/summary.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
cat $O/summary.txt
