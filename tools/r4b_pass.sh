mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
(timeout 200 python probes/conv_rec2_ab.py --zeros --shapes 0,4,1 2>&1 | grep -v amdgpu.ids) > $O/r4b_ab_zeros.log 2>&1; cat $O/r4b_ab_zeros.log
cd /tmp
(timeout 120 rocprofv3 --list-avail 2>&1 | grep -i -E "ICACHE|IFETCH|SQC_|SQ_INST_LEVEL|SQ_WAIT|SQ_BUSY|SQ_LEVEL" | head -80) > $O/r4b_counters_avail.log 2>&1
(timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/sq_a_r4b -o p -- python $R/probes/conv_rec2_ab.py --shapes 0,4,1 2>&1 | tail -4) > $O/sq_a_r4b.log 2>&1
(timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq_b_r4b -o p -- python $R/probes/conv_rec2_ab.py --shapes 0,4,1 2>&1 | tail -4) > $O/sq_b_r4b.log 2>&1
(timeout 300 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES GRBM_GUI_ACTIVE --output-format csv -d $O/sq_c_r4b -o p -- python $R/probes/conv_rec2_ab.py --shapes 0,4,1 2>&1 | tail -4) > $O/sq_c_r4b.log 2>&1
cd $R; python tools/pmc_sq.py $O/pmc_sq_summary_r4b.json $O/sq_a_r4b $O/sq_b_r4b $O/sq_c_r4b 2>&1 | tee $O/pmc_sq_r4b.log; tail -3 $O/sq_a_r4b.log $O/sq_c_r4b.log; rm -rf $O/sq_a_r4b $O/sq_b_r4b $O/sq_c_r4b
