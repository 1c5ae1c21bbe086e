# round-1 GPU pass: parity tests, smoke, bench, rocprofv3 kernel trace of the bench command
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
(timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -60) > gpurun_out/pytest_gpu.log 2>&1
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5) > gpurun_out/smoke.log 2>&1
(timeout 900 python bench.py --steps 1 --warmup 1 2>&1 | tail -5) > gpurun_out/bench1.log 2>&1
cd /tmp
(timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1 -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -5) > $R/gpurun_out/rocprof.log 2>&1
cd $R
find gpurun_out/prof_r1 -name "*.db" -delete 2>/dev/null
ls -la gpurun_out/prof_r1/* | head
tail -5 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/smoke.log; tail -2 gpurun_out/bench1.log
