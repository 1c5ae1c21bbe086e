# One GPU pass: parity suite, smoke, conv probe, short bench, rocprofv3 kernel trace.   usage: bash tools/gpu_pass.sh <tag> [what...]
# what: tests probe bench prof smoke large   (default: all but `prof`)
TAG=${1:-r2}; shift
WHAT=${*:-"tests smoke probe bench"}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
for w in $WHAT; do
  case $w in
    tests) (timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x --deselect tests/test_gpu_vae_large.py 2>&1 | tail -40) > $O/pytest_gpu_$TAG.log 2>&1; tail -8 $O/pytest_gpu_$TAG.log;;
    attnt) (timeout 600 python -m pytest tests/test_gpu_vae.py tests/test_gpu_seqpar.py tests/test_gpu_vae_large.py -m gpu -q --tb=short -p no:cacheprovider -k "attention or attn or seqpar or two_bench" 2>&1 | tail -30) > $O/pytest_attn_$TAG.log 2>&1; tail -12 $O/pytest_attn_$TAG.log;;
    new) (timeout 900 python -m pytest tests/test_gpu_ext.py tests/test_gpu_shard.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -60) > $O/pytest_new_$TAG.log 2>&1; tail -40 $O/pytest_new_$TAG.log;;
    rec) (timeout 900 python -m pytest tests/test_gpu_rec.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -60) > $O/pytest_rec_$TAG.log 2>&1; tail -30 $O/pytest_rec_$TAG.log;;
    large) (timeout 600 python -m pytest tests/test_gpu_vae_large.py -m gpu -q --tb=short -p no:cacheprovider -s --durations=10 2>&1 | grep -v "Tiled VAE\|amdgpu.ids" | tail -40) > $O/pytest_large_$TAG.log 2>&1; tail -15 $O/pytest_large_$TAG.log;;
    smoke) (timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2) > $O/smoke_$TAG.log 2>&1; cat $O/smoke_$TAG.log;;
    probe) (timeout 300 python probes/conv_probe.py 2>&1 | grep -v amdgpu.ids) > $O/conv_probe_$TAG.log 2>&1; cat $O/conv_probe_$TAG.log;;
    attn) (timeout 300 python probes/attn_probe.py 30000 77284 2>&1 | grep -v amdgpu.ids) > $O/attn_probe_$TAG.log 2>&1; cat $O/attn_probe_$TAG.log;;
    bench) (timeout 900 python bench.py --steps 2 --warmup 1 2>&1 | tail -1) > $O/bench_$TAG.json 2>&1; cut -c1-3000 $O/bench_$TAG.json;;
    benchq) (timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32-pass 2>&1 | tail -1) > $O/benchq_$TAG.json 2>&1; cut -c1-2500 $O/benchq_$TAG.json;;
    prof) cd /tmp; (timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-f32-pass 2>&1 | tail -3) > $O/rocprof_$TAG.log 2>&1; cd $R
          find $O/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$TAG.csv \; ; head -12 $O/kernel_stats_$TAG.csv | cut -c1-200;;
    sq) cd /tmp   # SQ / GRBM counters of the conv + attention probes (counters in their own runs, no tracing)
        (timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/sq_a_$TAG -o p -- python $R/probes/conv_probe.py --shapes 2,1,8 2>&1 | tail -4) > $O/sq_a_$TAG.log 2>&1
        (timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq_b_$TAG -o p -- python $R/probes/conv_probe.py --shapes 2,1,8 2>&1 | tail -4) > $O/sq_b_$TAG.log 2>&1
        cd $R; python tools/pmc_sq.py $O/pmc_sq_summary_$TAG.json $O/sq_a_$TAG $O/sq_b_$TAG 2>&1 | tee $O/pmc_sq_$TAG.log; tail -3 $O/sq_a_$TAG.log; rm -rf $O/sq_a_$TAG $O/sq_b_$TAG;;
    pmc) cd /tmp
         (timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$TAG -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass --no-f32-pass 2>&1 | tail -3) > $O/pmc_fetch_$TAG.log 2>&1
         (timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$TAG -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass --no-f32-pass 2>&1 | tail -3) > $O/pmc_write_$TAG.log 2>&1
         cd $R; (python tools/pmc_summary.py $O/pmc_fetch_$TAG $O/pmc_write_$TAG $O/pmc_hbm_summary_$TAG.json) > $O/pmc_summary_$TAG.log 2>&1; cat $O/pmc_summary_$TAG.log; rm -rf $O/pmc_fetch_$TAG $O/pmc_write_$TAG;;
  esac
done
find gpurun_out -name "*.db" -delete 2>/dev/null
find gpurun_out -name "*kernel_trace.csv" -size +8M -exec gzip -f {} \;
