# Quick regression check after a host-side refactor: GPU parity suite + one bench step
TAG=${1:-chk}
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
(timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -30) > $O/pytest_gpu_$TAG.log 2>&1
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2) > $O/smoke_$TAG.log 2>&1
(timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-600) > $O/bench_$TAG.log 2>&1
tail -5 $O/pytest_gpu_$TAG.log; cat $O/smoke_$TAG.log; cat $O/bench_$TAG.log
