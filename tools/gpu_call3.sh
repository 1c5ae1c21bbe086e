# Call 3: parity (incl. 2/3-process sequence-parallel estimator on one GPU, 1x1 split-bf16 conv, DMA attention), probes, bench + kernel trace
TAG=${1:-r1f}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
(timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -60) > $O/pytest_gpu_$TAG.log 2>&1
(MDTILE_ATTN_DMA=0 timeout 300 python -m pytest tests/test_gpu_vae.py -m gpu -q --tb=short -p no:cacheprovider -k "attention or attn" 2>&1 | tail -8) > $O/pytest_attn_nodma_$TAG.log 2>&1
(timeout 200 python probes/attn_probe.py 30000 71168 77284 2>&1 | grep -v amdgpu.ids) > $O/attn_probe_$TAG.log 2>&1
(MDTILE_ATTN_DMA=0 timeout 200 python probes/attn_probe.py 30000 77284 2>&1 | grep -v amdgpu.ids | sed 's/^/nodma /') >> $O/attn_probe_$TAG.log 2>&1
(timeout 200 python probes/conv_probe.py --no-exact 2>&1 | grep -v amdgpu.ids) > $O/conv_probe_$TAG.log 2>&1
(timeout 600 python bench.py --steps 1 --warmup 1 2>&1 | tail -3) > $O/bench_$TAG.log 2>&1
cd /tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | tail -3) > $O/rocprof_$TAG.log 2>&1
cd $R
find gpurun_out -name "*.db" -delete 2>/dev/null
find gpurun_out -name "*kernel_trace.csv" -size +8M -exec gzip -f {} \;
tail -25 $O/pytest_gpu_$TAG.log; tail -3 $O/pytest_attn_nodma_$TAG.log; cat $O/attn_probe_$TAG.log; cat $O/conv_probe_$TAG.log; tail -1 $O/bench_$TAG.log | cut -c1-1800
