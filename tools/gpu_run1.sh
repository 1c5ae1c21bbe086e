mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -150) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -20) > gpurun_out/smoke.log 2>&1
(timeout 600 python bench.py --steps 1 --warmup 1 2>&1 | tail -30) > gpurun_out/bench1.log 2>&1
tail -5 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/smoke.log; tail -2 gpurun_out/bench1.log
