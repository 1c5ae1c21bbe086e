# Final pass of a round: parity, smoke, HBM PMC passes -> summary, bench (driver defaults) with traffic filled in, kernel trace,
# and a functional run of the N = 2 bench flow on one GPU.   usage: bash tools/gpu_final.sh [tag]
TAG=${1:-r1h}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -40) > $O/pytest_gpu_$TAG.log 2>&1
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3) > $O/smoke_$TAG.log 2>&1
(timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline --debug-single-device 2>&1 | tail -15) > $O/bench_n2_debug_$TAG.log 2>&1
cd /tmp
(timeout 600 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_fetch_$TAG -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass 2>&1 | tail -3) > $O/pmc_fetch_$TAG.log 2>&1
(timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$TAG -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass 2>&1 | tail -3) > $O/pmc_write_$TAG.log 2>&1
cd $R
(python tools/pmc_summary.py $O/pmc_fetch_$TAG $O/pmc_write_$TAG $O/pmc_hbm_summary_$TAG.json) > $O/pmc_summary_$TAG.log 2>&1
(MDTILE_PMC_SUMMARY=$O/pmc_hbm_summary_$TAG.json timeout 900 python bench.py 2>&1 | tail -3) > $O/bench_$TAG.log 2>&1
cd /tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | tail -3) > $O/rocprof_$TAG.log 2>&1
cd $R
(timeout 300 python probes/encode_probe.py 2>&1 | grep -v amdgpu.ids) > $O/encode_probe_$TAG.log 2>&1
(timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile-pass --slow-vae 2>&1 | tail -1 | cut -c1-900 | sed "s/^/slow-mode GroupNorm: /") > $O/bench_extra_$TAG.log 2>&1
(timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile-pass --method mod 2>&1 | tail -1 | cut -c1-900 | sed "s/^/Mixture of Diffusers blend: /") >> $O/bench_extra_$TAG.log 2>&1
(timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile-pass --latent 512 --tile 96 --overlap 48 --method mod 2>&1 | tail -1 | cut -c1-900 | sed "s/^/cfg3 4096x4096 MoD 96\/48 + tiled decode: /") >> $O/bench_extra_$TAG.log 2>&1
(timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile-pass --latent 256 --tile 96 --overlap 48 --no-vae 2>&1 | tail -1 | cut -c1-900 | sed "s/^/cfg2 2048x2048 MD 96\/48 blend only: /") >> $O/bench_extra_$TAG.log 2>&1
(timeout 200 python probes/attn_probe.py 30000 71168 77284 2>&1 | grep -v amdgpu.ids) > $O/attn_probe_$TAG.log 2>&1
(timeout 200 python probes/conv_probe.py --no-exact 2>&1 | grep -v amdgpu.ids) > $O/conv_probe_$TAG.log 2>&1
find gpurun_out -name "*.db" -delete 2>/dev/null
rm -rf $O/pmc_fetch_$TAG $O/pmc_write_$TAG
find gpurun_out -name "*kernel_trace.csv" -size +8M -exec gzip -f {} \;
tail -6 $O/pytest_gpu_$TAG.log; tail -2 $O/smoke_$TAG.log; tail -8 $O/bench_n2_debug_$TAG.log | cut -c1-600; cat $O/pmc_summary_$TAG.log; cat $O/encode_probe_$TAG.log; cat $O/bench_extra_$TAG.log; tail -1 $O/bench_$TAG.log | cut -c1-2500
