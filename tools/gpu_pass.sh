# GPU pass: parity tests, smoke, bench, rocprofv3 kernel trace (csv stats) + HBM PMC passes of the bench command.
# usage (on the GPU box, from the repo root):  bash tools/gpu_pass.sh [tag] [skip_tests]
TAG=${1:-r1}
SKIP_TESTS=${2:-0}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
if [ "$SKIP_TESTS" = "0" ]; then
  (timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -40) > gpurun_out/pytest_gpu_$TAG.log 2>&1
  (timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5) > gpurun_out/smoke_$TAG.log 2>&1
fi
(timeout 900 python bench.py --steps 1 --warmup 1 2>&1 | tail -5) > gpurun_out/bench_$TAG.log 2>&1
cd /tmp
(timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | tail -5) > $R/gpurun_out/rocprof_$TAG.log 2>&1
(timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$TAG -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass 2>&1 | tail -5) > $R/gpurun_out/pmc_fetch_$TAG.log 2>&1
(timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_$TAG -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass 2>&1 | tail -5) > $R/gpurun_out/pmc_write_$TAG.log 2>&1
cd $R
find gpurun_out -name "*.db" -delete 2>/dev/null
# the raw per-dispatch traces are large; keep them zipped (<= 64 MiB merges back), the stats csv stays readable
find gpurun_out -name "*kernel_trace.csv" -size +8M -exec gzip -f {} \;
find gpurun_out -name "*counter_collection.csv" -size +8M -exec gzip -f {} \;
find gpurun_out -type f | head -60; du -sh gpurun_out
tail -5 gpurun_out/pytest_gpu_$TAG.log; tail -2 gpurun_out/smoke_$TAG.log; tail -2 gpurun_out/bench_$TAG.log
