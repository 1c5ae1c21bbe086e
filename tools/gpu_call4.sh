# Call 4: encoder parity, conv kernel variants (64-cout blocks / weight DMA / two blocks per CU): parity + probe + bench each
TAG=${1:-r1g}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
(timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -60) > $O/pytest_gpu_$TAG.log 2>&1
for V in MDTILE_CONV_MT=2 MDTILE_CONV_WDMA=1 MDTILE_CONV_OCC2=1; do
  (env $V timeout 300 python -m pytest tests/test_gpu_vae.py -m gpu -q --tb=short -p no:cacheprovider -k "conv2d or fused or full_width_decoder" 2>&1 | tail -12 | sed "s/^/$V /") >> $O/pytest_variants_$TAG.log 2>&1
done
(timeout 200 python probes/conv_probe.py --no-exact --shapes 0,2,4,5,7,8 2>&1 | grep -v amdgpu.ids | sed "s/^/default /") > $O/conv_probe_$TAG.log 2>&1
for V in MDTILE_CONV_MT=2 MDTILE_CONV_WDMA=1 MDTILE_CONV_OCC2=1; do
  (env $V timeout 200 python probes/conv_probe.py --no-exact --shapes 0,2,4,5,7,8 2>&1 | grep -v amdgpu.ids | sed "s/^/$V /") >> $O/conv_probe_$TAG.log 2>&1
done
for V in MDTILE_CONV_WDMA=1 MDTILE_CONV_OCC2=1 MDTILE_CONV_MT=2; do
  (env $V timeout 400 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | tail -1 | sed "s/^/$V /") >> $O/bench_variants_$TAG.log 2>&1
done
tail -30 $O/pytest_gpu_$TAG.log; cat $O/pytest_variants_$TAG.log | grep -v "^\S* *$" | tail -20; cat $O/conv_probe_$TAG.log; cut -c1-1500 $O/bench_variants_$TAG.log
