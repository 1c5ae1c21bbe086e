# Call 7: MFMA order variants (MDTILE_CONV_ORD): parity + probe
TAG=${1:-r1l}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
(MDTILE_CONV_ORD=2 timeout 400 python -m pytest tests/test_gpu_vae.py -m gpu -q --tb=short -p no:cacheprovider -k "conv2d or fused or full_width_decoder" 2>&1 | tail -8) > $O/pytest_ord2_$TAG.log 2>&1
(timeout 200 python probes/conv_probe.py --no-exact --shapes 0,2,4,5,7,8 2>&1 | grep -v amdgpu.ids | sed "s/^/default /") > $O/conv_probe_$TAG.log 2>&1
(MDTILE_CONV_ORD=2 timeout 200 python probes/conv_probe.py --no-exact --shapes 0,2,4,5,7,8 2>&1 | grep -v amdgpu.ids | sed "s/^/ORD2 /") >> $O/conv_probe_$TAG.log 2>&1
(MDTILE_CONV_ORD=2 MDTILE_CONV_OCC2=0 timeout 200 python probes/conv_probe.py --no-exact --shapes 0,2,4,5,7,8 2>&1 | grep -v amdgpu.ids | sed "s/^/TH16+ORD2 /") >> $O/conv_probe_$TAG.log 2>&1
(MDTILE_CONV_ORD=1 timeout 200 python probes/conv_probe.py --no-exact --shapes 2,4 2>&1 | grep -v amdgpu.ids | sed "s/^/ORD1 /") >> $O/conv_probe_$TAG.log 2>&1
tail -3 $O/pytest_ord2_$TAG.log; cat $O/conv_probe_$TAG.log
