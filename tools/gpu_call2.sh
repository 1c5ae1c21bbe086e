# Call 2: parity of the new kernels (sub-pixel upconv, fused GN conv, 16-row conv), variant probes, SQ counters on the conv
# and attention kernels, bench.   usage: bash tools/gpu_call2.sh [tag]
TAG=${1:-r1e}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
(timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -40) > $O/pytest_gpu_$TAG.log 2>&1
(MDTILE_CONV_TH=16 timeout 300 python -m pytest tests/test_gpu_vae.py -m gpu -q --tb=short -p no:cacheprovider -k "conv2d or fused or full_width" 2>&1 | tail -15) > $O/pytest_th16_$TAG.log 2>&1
(timeout 200 python probes/conv_probe.py --no-exact 2>&1 | grep -v amdgpu.ids) > $O/conv_probe_$TAG.log 2>&1
(MDTILE_CONV_TH=16 timeout 200 python probes/conv_probe.py --no-exact --shapes 0,2,4,5,7,8 2>&1 | grep -v amdgpu.ids) >> $O/conv_probe_$TAG.log 2>&1
(MDTILE_UPCONV=direct timeout 200 python probes/conv_probe.py --no-exact --shapes 1,3,6 2>&1 | grep -v amdgpu.ids) >> $O/conv_probe_$TAG.log 2>&1
(timeout 200 python probes/attn_probe.py 2>&1 | grep -v amdgpu.ids) > $O/attn_probe_$TAG.log 2>&1
cd /tmp
(timeout 60 rocprofv3 -L 2>&1 | grep -o "SQ_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|GRBM_[A-Z0-9_]*\|TA_[A-Z0-9_]*" | sort -u) > $O/pmc_list.txt 2>&1
PA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
PB="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"
(timeout 300 rocprofv3 --pmc $PA --kernel-trace --output-format csv -d $O/pmc_conv_a -o p -- python $R/probes/conv_probe.py --no-exact --shapes 2,1 2>&1 | tail -3) > $O/pmc_conv_a.log 2>&1
(timeout 300 rocprofv3 --pmc $PB --kernel-trace --output-format csv -d $O/pmc_conv_b -o p -- python $R/probes/conv_probe.py --no-exact --shapes 2,1 2>&1 | tail -3) > $O/pmc_conv_b.log 2>&1
(timeout 300 rocprofv3 --pmc $PA --kernel-trace --output-format csv -d $O/pmc_attn_a -o p -- python $R/probes/attn_probe.py --quick 2>&1 | tail -3) > $O/pmc_attn_a.log 2>&1
cd $R
(timeout 600 python bench.py --steps 1 --warmup 1 2>&1 | tail -3) > $O/bench_$TAG.log 2>&1
(MDTILE_FUSE_GN=0 timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | tail -3) > $O/bench_nofuse_$TAG.log 2>&1
find gpurun_out -name "*.db" -delete 2>/dev/null
find gpurun_out -name "*kernel_trace.csv" -size +8M -exec gzip -f {} \;
find gpurun_out -name "*counter_collection.csv" -size +8M -exec gzip -f {} \;
tail -4 $O/pytest_gpu_$TAG.log; tail -3 $O/pytest_th16_$TAG.log; cat $O/conv_probe_$TAG.log; cat $O/attn_probe_$TAG.log; tail -2 $O/bench_$TAG.log | cut -c1-1500; tail -1 $O/bench_nofuse_$TAG.log | cut -c1-400
