# GPU pass: parity tests, bench, rocprofv3 kernel trace (csv) + HBM PMC passes of the bench command
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
(timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -40) > gpurun_out/pytest_gpu.log 2>&1
(timeout 900 python bench.py --steps 1 --warmup 1 2>&1 | tail -5) > gpurun_out/bench1.log 2>&1
cd /tmp
(timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1 -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -5) > $R/gpurun_out/rocprof.log 2>&1
(timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | tail -5) > $R/gpurun_out/pmc_fetch.log 2>&1
(timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | tail -5) > $R/gpurun_out/pmc_write.log 2>&1
cd $R
find gpurun_out -name "*.db" -delete 2>/dev/null
find gpurun_out -type f | head -40; du -sh gpurun_out
tail -5 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/bench1.log
