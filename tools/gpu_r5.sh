# Round-5 GPU pass.   usage: bash tools/gpu_r5.sh <tag> [what...]     (through gpurun; artefacts land in gpurun_out/)
# what: every step of tools/gpu_r3.sh (tests | smoke | bench | benchq | prof | pmc | ...) plus
#   sqdrip   SQ / GRBM counters of probes/conv_drip_ab.py (one-block record conv vs the dripped-epilogue conv, 128 -> 128 at 2224^2) -> pmc_sq_drip_<tag>.json
#   dripab   the same-process A/B of the two kernels on the decoder's shapes (+ K loops alone and the slot ablations on the PROBES twin)
#   encode   bench.py --encode (the ENCODE companion line)
#   stress   tests/test_gpu_vae_stress.py -s (the numbers it prints)
#   micro    probes/simd_map_probe + probes/mfma_valu_overlap_probe
#   sqbench  SQ / GRBM counters of the BENCH command itself (the driver line's launch mix; two --pmc passes, no tracing) -> pmc_sq_summary_<tag>.json
#   slowprof rocprofv3 --kernel-trace --stats of one slow-mode step (bench.py --slow-vae) -> kernel_stats_slow_<tag>.csv
#   slow     bench.py --slow-vae with the statistics from the conv epilogues and (MDTILE_SLOW_STATS=0) with a statistics pass per norm, twice each
TAG=${1:-r5}; shift
WHAT=${*:-"tests smoke bench"}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
for w in $WHAT; do
  case $w in
    sqdrip) cd /tmp
        (timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/sqd_a_$TAG -o p -- python $R/probes/conv_drip_ab.py --shapes 0 2>&1 | tail -2) > $O/sqd_a_$TAG.log 2>&1
        (timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sqd_b_$TAG -o p -- python $R/probes/conv_drip_ab.py --shapes 0 2>&1 | tail -2) > $O/sqd_b_$TAG.log 2>&1
        cd $R; python tools/pmc_sq.py $O/pmc_sq_drip_$TAG.json $O/sqd_a_$TAG $O/sqd_b_$TAG 2>&1 | tail -30 | tee $O/pmc_sq_drip_$TAG.log; tail -2 $O/sqd_a_$TAG.log; rm -rf $O/sqd_a_$TAG $O/sqd_b_$TAG;;
    dripab) (timeout 400 python probes/conv_drip_ab.py 2>&1 | grep -v amdgpu.ids) > $O/conv_drip_ab_$TAG.log 2>&1
        (timeout 300 python probes/conv_drip_ab.py --b 3 --shapes 5,4 2>&1 | grep -v amdgpu.ids) >> $O/conv_drip_ab_$TAG.log 2>&1
        for d in 1 33 97 2 4 6 12 14 16 30; do echo "MDTILE_REC_DBG=$d (PROBES twin)" >> $O/conv_drip_ablation_$TAG.log; (timeout 200 python probes/conv_drip_ab.py --shapes 0 --dbg $d 2>&1 | grep -v "amdgpu.ids\|probes\]") >> $O/conv_drip_ablation_$TAG.log 2>&1; done
        cat $O/conv_drip_ab_$TAG.log;;
    encode) (timeout 900 python bench.py --encode --steps 2 --warmup 1 2>&1 | tail -1) > $O/bench_encode_$TAG.json 2>&1; cut -c1-3000 $O/bench_encode_$TAG.json;;
    stress) (timeout 900 python -m pytest tests/test_gpu_vae_stress.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep "stress\|attention at\|passed\|failed") > $O/pytest_stress_$TAG.log 2>&1; cat $O/pytest_stress_$TAG.log;;
    sqbench) cd /tmp; BF="--steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass --no-f32-pass --no-oracle-pass --no-whole-tile-pass --no-stress-pass --no-companions"
        (timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/sqm_a_$TAG -o p -- python $R/bench.py $BF 2>&1 | tail -2) > $O/sqm_a_$TAG.log 2>&1
        (timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sqm_b_$TAG -o p -- python $R/bench.py $BF 2>&1 | tail -2) > $O/sqm_b_$TAG.log 2>&1
        cd $R; python tools/pmc_sq.py $O/pmc_sq_summary_$TAG.json $O/sqm_a_$TAG $O/sqm_b_$TAG 2>&1 | grep "k_conv3x3_rec\|k_attn_bf16x3\|k_upconv_rec\|kernel" | head -20 | tee $O/pmc_sq_$TAG.log; tail -2 $O/sqm_a_$TAG.log; rm -rf $O/sqm_a_$TAG $O/sqm_b_$TAG;;
    slowprof) cd /tmp; (timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profs_$TAG -o bench -- python $R/bench.py --slow-vae --steps 1 --warmup 0 --no-cpu-baseline --no-f32-pass --no-oracle-pass --no-whole-tile-pass --no-stress-pass --no-companions 2>&1 | tail -3) > $O/rocprof_slow_$TAG.log 2>&1; cd $R
        find $O/profs_$TAG -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_slow_$TAG.csv \; ; head -16 $O/kernel_stats_slow_$TAG.csv | cut -c1-200; rm -rf $O/profs_$TAG;;
    slow) : > $O/slow_ab_$TAG.log
        for v in 1 0 1 0; do MDTILE_SLOW_STATS=$v timeout 600 python bench.py --slow-vae --steps 2 --warmup 1 --no-oracle-pass --no-stress-pass --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('MDTILE_SLOW_STATS=$v  slow-mode 8K step', d['ms_per_step'], 'ms')" >> $O/slow_ab_$TAG.log 2>&1; done
        cat $O/slow_ab_$TAG.log;;
    micro) (probes/simd_map_probe; probes/mfma_valu_overlap_probe) > $O/micro_$TAG.log 2>&1; cat $O/micro_$TAG.log;;
    *) bash $R/tools/gpu_r3.sh $TAG $w;;
  esac
done
find gpurun_out -name "*.db" -delete 2>/dev/null
