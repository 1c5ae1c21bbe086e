TAG=$1
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; export TMPDIR=/tmp; cd /tmp
(timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/sqa_$TAG -o p -- python $R/probes/attn_probe.py 77284 --quick 2>&1 | tail -3) > $O/sqa_$TAG.log 2>&1
(timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sqb_$TAG -o p -- python $R/probes/attn_probe.py 77284 --quick 2>&1 | tail -3) > $O/sqb_$TAG.log 2>&1
cd $R; python tools/pmc_sq.py $O/pmc_sq_attn_$TAG.json $O/sqa_$TAG $O/sqb_$TAG 2>&1 | tee $O/pmc_sq_attn_$TAG.log; rm -rf $O/sqa_$TAG $O/sqb_$TAG
