# Round-3 GPU pass.   usage: bash tools/gpu_r3.sh <tag> [what...]     (through gpurun; artefacts land in gpurun_out/)
# what: tests (whole -m gpu suite) | new (this round's new tests) | smoke | bench | benchq | prof | pmc | rccl | attn | conv | blend | probe:<cmd>
TAG=${1:-r3}; shift
WHAT=${*:-"tests smoke bench"}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
for w in $WHAT; do
  case $w in
    tests) (timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -60) > $O/pytest_gpu_$TAG.log 2>&1; tail -15 $O/pytest_gpu_$TAG.log;;
    new) (timeout 900 python -m pytest tests/test_gpu_shard.py tests/test_md_entry_path.py tests/test_gpu_vae_large.py -m gpu -q --tb=short -p no:cacheprovider -s --durations=8 2>&1 | grep -v "Tiled VAE\|amdgpu.ids\|Sampling" | tail -60) > $O/pytest_new_$TAG.log 2>&1; tail -30 $O/pytest_new_$TAG.log;;
    smoke) (timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2) > $O/smoke_$TAG.log 2>&1; cat $O/smoke_$TAG.log;;
    bench) (timeout 1200 python bench.py --steps 2 --warmup 1 2>&1 | tail -1) > $O/bench_$TAG.json 2>&1; cut -c1-6000 $O/bench_$TAG.json;;
    benchq) (timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32-pass --no-oracle-pass --no-whole-tile-pass --no-stress-pass --no-companions 2>&1 | tail -1) > $O/benchq_$TAG.json 2>&1; cut -c1-4000 $O/benchq_$TAG.json;;
    prof) cd /tmp; (timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-f32-pass --no-oracle-pass --no-whole-tile-pass --no-stress-pass --no-companions 2>&1 | tail -3) > $O/rocprof_$TAG.log 2>&1; cd $R
          find $O/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$TAG.csv \; ; head -14 $O/kernel_stats_$TAG.csv | cut -c1-200; rm -rf $O/prof_$TAG;;
    pmc) cd /tmp
         (timeout 600 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_fetch_$TAG -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass --no-f32-pass --no-oracle-pass --no-whole-tile-pass --no-stress-pass --no-companions 2>&1 | tail -3) > $O/pmc_fetch_$TAG.log 2>&1
         (timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$TAG -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass --no-f32-pass --no-oracle-pass --no-whole-tile-pass --no-stress-pass --no-companions 2>&1 | tail -3) > $O/pmc_write_$TAG.log 2>&1
         cd $R; (python tools/pmc_summary.py $O/pmc_fetch_$TAG $O/pmc_write_$TAG $O/pmc_hbm_summary_$TAG.json) > $O/pmc_summary_$TAG.log 2>&1; cat $O/pmc_summary_$TAG.log; tail -2 $O/pmc_fetch_$TAG.log; rm -rf $O/pmc_fetch_$TAG $O/pmc_write_$TAG;;
    rccl) cd /tmp   # RCCL kernels of the 1-rank communicator tests, by name
          (timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rccl_$TAG -o rccl -- python -m pytest $R/tests/test_gpu_shard.py -m gpu -q -p no:cacheprovider -k "rccl_one_rank" 2>&1 | tail -5) > $O/rccl_$TAG.log 2>&1; cd $R
          find $O/rccl_$TAG -name "*kernel_stats.csv" -exec cp {} $O/rccl_kernel_stats_$TAG.csv \; ; cat $O/rccl_$TAG.log; cut -c1-160 $O/rccl_kernel_stats_$TAG.csv | head -20; rm -rf $O/rccl_$TAG;;
    dbg2) (timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --debug-single-device --steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass 2>&1 | grep -v "amdgpu.ids\|Tiled VAE\|Sampling" | tail -12) > $O/bench_n2_single_device_$TAG.log 2>&1; cut -c1-1800 $O/bench_n2_single_device_$TAG.log;;
    dbg8) (MDTILE_TILE_BATCH=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 8 --debug-single-device --steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass 2>&1 | grep -v "amdgpu.ids\|Tiled VAE\|Sampling" | tail -12) > $O/bench_n8_single_device_$TAG.log 2>&1; cut -c1-1800 $O/bench_n8_single_device_$TAG.log;;
    vae) (timeout 900 python -m pytest tests/test_gpu_vae.py tests/test_gpu_rec.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "Tiled VAE\|amdgpu.ids" | tail -40) > $O/pytest_vae_$TAG.log 2>&1; tail -25 $O/pytest_vae_$TAG.log;;
    c1x1) (timeout 300 python probes/conv1x1_probe.py 2>&1 | grep -v amdgpu.ids) > $O/conv1x1_probe_$TAG.log 2>&1; cat $O/conv1x1_probe_$TAG.log;;
    enc) (timeout 600 python probes/encode_probe.py 2>&1 | grep -v amdgpu.ids) > $O/encode_probe_$TAG.log 2>&1; cat $O/encode_probe_$TAG.log;;
    attn) (timeout 300 python probes/attn_probe.py 30000 77284 2>&1 | grep -v amdgpu.ids) > $O/attn_probe_$TAG.log 2>&1; cat $O/attn_probe_$TAG.log;;
    conv) (timeout 300 python probes/conv_probe.py 2>&1 | grep -v amdgpu.ids) > $O/conv_probe_$TAG.log 2>&1; cat $O/conv_probe_$TAG.log;;
    blend) (timeout 300 python probes/blend_ab.py 2>&1 | grep -v amdgpu.ids) > $O/blend_ab_$TAG.log 2>&1; cat $O/blend_ab_$TAG.log;;
    probe:*) cmd=${w#probe:}; (timeout 600 bash -c "${cmd//+/ }" 2>&1 | grep -v amdgpu.ids | tail -60) > $O/probe_$TAG.log 2>&1; cat $O/probe_$TAG.log;;
  esac
done
find gpurun_out -name "*.db" -delete 2>/dev/null
find gpurun_out -name "*kernel_trace.csv" -size +8M -exec gzip -f {} \;
