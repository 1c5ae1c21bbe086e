# Lean GPU pass: parity tests, smoke, bench, rocprofv3 kernel trace of the bench command (no PMC passes).
# usage (GPU box, repo root):  bash tools/gpu_lean.sh [tag] [extra pytest args]
TAG=${1:-r1}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
(timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -40) > gpurun_out/pytest_gpu_$TAG.log 2>&1
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5) > gpurun_out/smoke_$TAG.log 2>&1
(timeout 900 python bench.py --steps 1 --warmup 1 2>&1 | tail -5) > gpurun_out/bench_$TAG.log 2>&1
cd /tmp
(timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | tail -5) > $R/gpurun_out/rocprof_$TAG.log 2>&1
cd $R
find gpurun_out -name "*.db" -delete 2>/dev/null
find gpurun_out -name "*kernel_trace.csv" -size +8M -exec gzip -f {} \;
tail -5 gpurun_out/pytest_gpu_$TAG.log; tail -2 gpurun_out/smoke_$TAG.log; tail -2 gpurun_out/bench_$TAG.log
