# Call 6: 3-stage weight ring (MDTILE_CONV_W3=1): parity, probe vs default, bench
TAG=${1:-r1k}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
(MDTILE_CONV_W3=1 timeout 400 python -m pytest tests/test_gpu_vae.py -m gpu -q --tb=short -p no:cacheprovider -k "conv2d or fused or decode" 2>&1 | tail -15) > $O/pytest_w3_$TAG.log 2>&1
(timeout 200 python probes/conv_probe.py --no-exact --shapes 0,2,4,5,7,8 2>&1 | grep -v amdgpu.ids | sed "s/^/default /") > $O/conv_probe_$TAG.log 2>&1
(MDTILE_CONV_W3=1 timeout 200 python probes/conv_probe.py --no-exact --shapes 0,2,4,5,7,8 2>&1 | grep -v amdgpu.ids | sed "s/^/W3 /") >> $O/conv_probe_$TAG.log 2>&1
(MDTILE_CONV_OCC2=0 timeout 200 python probes/conv_probe.py --no-exact --shapes 0,2,4,5,7,8 2>&1 | grep -v amdgpu.ids | sed "s/^/TH16 /") >> $O/conv_probe_$TAG.log 2>&1
(MDTILE_CONV_W3=1 timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_w3_$TAG.log 2>&1
tail -4 $O/pytest_w3_$TAG.log; cat $O/conv_probe_$TAG.log; cut -c1-1700 $O/bench_w3_$TAG.log
