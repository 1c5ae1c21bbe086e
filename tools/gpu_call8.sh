# Call 8: SQ counters of the round's final conv / upconv / attention kernels (separate --pmc passes, kernel trace only)
TAG=${1:-r1n}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
PA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
PB="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_SALU"
cd /tmp
(timeout 300 rocprofv3 --pmc $PA --kernel-trace --output-format csv -d $O/pmc_conv_a -o p -- python $R/probes/conv_probe.py --no-exact --shapes 2,1,5 2>&1 | tail -3) > $O/pmc_conv_a.log 2>&1
(timeout 300 rocprofv3 --pmc $PB --kernel-trace --output-format csv -d $O/pmc_conv_b -o p -- python $R/probes/conv_probe.py --no-exact --shapes 2,1,5 2>&1 | tail -3) > $O/pmc_conv_b.log 2>&1
(timeout 300 rocprofv3 --pmc $PA --kernel-trace --output-format csv -d $O/pmc_attn_a -o p -- python $R/probes/attn_probe.py --quick 77284 2>&1 | tail -3) > $O/pmc_attn_a.log 2>&1
cd $R
tail -2 $O/pmc_conv_a.log $O/pmc_conv_b.log $O/pmc_attn_a.log
