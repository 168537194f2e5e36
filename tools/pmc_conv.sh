cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_conv; rm -rf $O; mkdir -p $O
ARGS="--K 192 --M 64 --bwd"
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 tools/kconv_only.py $ARGS > $O/kt.log 2>&1 &&
timeout -k 10 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/p1 -- python3 tools/kconv_only.py $ARGS > $O/p1.log 2>&1 &&
timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $O/p2 -- python3 tools/kconv_only.py $ARGS > $O/p2.log 2>&1 &&
timeout -k 10 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM --output-format csv -d $O/p3 -- python3 tools/kconv_only.py $ARGS > $O/p3.log 2>&1 &&
timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $O/p4 -- python3 tools/kconv_only.py $ARGS > $O/p4.log 2>&1
python tools/pmc_table.py $O/p1 $O/p2 $O/p3 $O/p4 > $O/table.txt 2>&1
python tools/trace_summary.py $(ls $O/kt/*/*kernel_trace.csv | head -1) > $O/kt.txt 2>&1
rm -rf $O/kt $O/p1 $O/p2 $O/p3 $O/p4
cat $O/kt.txt $O/table.txt; tail -3 $O/p4.log
