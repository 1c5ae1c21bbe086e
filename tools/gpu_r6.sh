# Round-6 GPU pass.   usage: bash tools/gpu_r6.sh <tag> [what...]     (through gpurun; artefacts land in gpurun_out/)
# what: every step of tools/gpu_r5.sh / gpu_r3.sh (tests | smoke | bench | benchq | prof | pmc | sqbench | encode | slow | ...) plus
#   r6new    this round's new GPU tests (bare / single-process bench forms, encoder stress parity, statistics calls, stream copy)
#   bare8    `python bench.py --gpus 8 --steps 1 --warmup 0` with NO launcher environment on this one-GPU box (rc, the JSON line)
#   sp8      `python bench.py --gpus 8 --single-process --steps 1 --warmup 1` (cuda:0 listed eight times)
#   blendab  probes/blend_r6_ab.py: the shipping blend kernel, the LDS-staged form and the stream-copy floor, cold and warm, same process
#   attnab   probes/attn_ab.py <other .so>: two builds of the attention kernel alternated in one process
#   contend  the co-residency probes (probes/contention_*.py);   convf32  probes/convf32_probe.py over the block shapes under test
#   chk:<pytest args with + for spaces>   an ad-hoc pytest selection
TAG=${1:-r6}; shift
WHAT=${*:-"tests smoke bench"}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
for w in $WHAT; do
  case $w in
    r6new) (timeout 2400 python -m pytest tests/test_gpu_shard.py tests/test_gpu_vae_stress.py tests/test_gpu_conv_stats.py tests/test_gpu_blend.py -m gpu -q --tb=short -p no:cacheprovider -s --durations=10 2>&1 | grep -v "Tiled VAE\|amdgpu.ids\|Sampling" | tail -70) > $O/pytest_r6new_$TAG.log 2>&1; tail -45 $O/pytest_r6new_$TAG.log;;
    bare8) (timeout 1500 python bench.py --gpus 8 --steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass > $O/bench_bare8_$TAG.json 2> $O/bench_bare8_$TAG.err; echo "rc=$?" >> $O/bench_bare8_$TAG.err); tail -3 $O/bench_bare8_$TAG.err | cut -c1-300; cut -c1-2500 $O/bench_bare8_$TAG.json;;
    sp8) (timeout 1500 python bench.py --gpus 8 --single-process --steps 1 --warmup 1 > $O/bench_sp8_$TAG.json 2> $O/bench_sp8_$TAG.err; echo "rc=$?" >> $O/bench_sp8_$TAG.err); tail -3 $O/bench_sp8_$TAG.err | cut -c1-300; cut -c1-2500 $O/bench_sp8_$TAG.json;;
    blendab) (timeout 900 python probes/blend_r6_ab.py 2>&1 | grep -v amdgpu.ids) > $O/blend_r6_ab_$TAG.log 2>&1
        if [ -f probes/_ab/libmdtile_r5_blend.so ]; then (MDTILE_AB_LIB=probes/_ab/libmdtile_r5_blend.so timeout 600 python probes/blend_r6_ab.py 2>&1 | grep -v amdgpu.ids) >> $O/blend_r6_ab_$TAG.log 2>&1; fi
        cat $O/blend_r6_ab_$TAG.log;;
    attnab) (timeout 900 python probes/attn_ab.py probes/_ab/libmdtile_attn_base.so 2>&1 | grep -v amdgpu.ids) > $O/attn_ab_$TAG.log 2>&1; cat $O/attn_ab_$TAG.log;;
    sqattn) # SQ / GRBM counters of the attention kernel at T = 77 284: the shipping kernel and the rejected round-6 variant (DMA pieces between MFMA pairs)
        for v in shipping interleaved; do
          if [ $v = interleaved ]; then export MDTILE_AB_LIB=$R/probes/_ab/libmdtile_attn_interleaved.so; else unset MDTILE_AB_LIB; fi
          cd /tmp
          (timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/sqa_$TAG -o p -- python $R/probes/attn_probe.py 77284 --quick 2>&1 | tail -3) > $O/sqa_$TAG.log 2>&1
          (timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sqb_$TAG -o p -- python $R/probes/attn_probe.py 77284 --quick 2>&1 | tail -3) > $O/sqb_$TAG.log 2>&1
          cd $R; python tools/pmc_sq.py $O/pmc_sq_attn_${v}_$TAG.json $O/sqa_$TAG $O/sqb_$TAG 2>&1 | grep "k_attn_bf16x3\|kernel" | head -6 | tee $O/pmc_sq_attn_${v}_$TAG.log; rm -rf $O/sqa_$TAG $O/sqb_$TAG
        done; unset MDTILE_AB_LIB;;
    c1x1mt) : > $O/conv1x1_mt_ab_$TAG.log      # 1x1 streaming conv: 256-cout x 256-px blocks (default for cout % 256 == 0) vs 128-cout x 512-px blocks
        for b in 1 4; do for mt in 8 4; do echo "PROBE_B=$b MDTILE_C1X1_MT=$mt" >> $O/conv1x1_mt_ab_$TAG.log
          (PROBE_B=$b MDTILE_C1X1_MT=$mt timeout 300 python probes/conv1x1_probe.py 2>&1 | grep "^1x1") >> $O/conv1x1_mt_ab_$TAG.log 2>&1; done; done
        cat $O/conv1x1_mt_ab_$TAG.log;;
    sqc1x1) cd /tmp      # SQ / GRBM counters of the 1x1 streaming conv on the decode's shapes (4 stacked tiles)
        (PROBE_B=4 timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/sqa_$TAG -o p -- python $R/probes/conv1x1_probe.py 2>&1 | tail -3) > $O/sqa_$TAG.log 2>&1
        (PROBE_B=4 timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sqb_$TAG -o p -- python $R/probes/conv1x1_probe.py 2>&1 | tail -3) > $O/sqb_$TAG.log 2>&1
        cd $R; python tools/pmc_sq.py $O/pmc_sq_c1x1_$TAG.json $O/sqa_$TAG $O/sqb_$TAG 2>&1 | grep "conv1x1\|kernel" | head -8; rm -rf $O/sqa_$TAG $O/sqb_$TAG;;
    c1x1ab) : > $O/conv1x1_ab_$TAG.log      # 1x1 streaming conv: this tree vs probes/_ab/libmdtile_c1x1_base.so (before the cooperative split), twice each
        for rep in 1 2; do for b in 1 4; do for lib in base tree; do echo "PROBE_B=$b $lib" >> $O/conv1x1_ab_$TAG.log
          if [ $lib = base ]; then export MDTILE_AB_LIB=$R/probes/_ab/libmdtile_c1x1_base.so; else unset MDTILE_AB_LIB; fi
          (PROBE_B=$b timeout 300 python probes/conv1x1_probe.py 2>&1 | grep "^1x1") >> $O/conv1x1_ab_$TAG.log 2>&1; done; done; done; unset MDTILE_AB_LIB
        cat $O/conv1x1_ab_$TAG.log;;
    contend) # conv_in / the blend / the whole decode sharing the GPU with other processes' and streams' kernels (profiles/r6j)
        { for form in "" 1; do for mode in "mix:handover" "mix" "same:handover"; do echo "=== MDTILE_FEWCIN_FORM=$form, $mode"
            MDTILE_FEWCIN_FORM=$form timeout 300 python probes/contention_fewcin.py 4 100 "$mode" 2>&1 | grep -v "amdgpu\|failure\|wrong value\|^\[probes\]" | tail -1; done; done
          timeout 300 python probes/contention_regkeep.py 4 100 mix:handover 2>&1 | grep -v "amdgpu\|^\[probes\]"
          timeout 600 python probes/contention_blend.py 4 200 mix:handover 2>&1 | grep -v "amdgpu\|^\[probes\]"
          timeout 900 python probes/contention_determinism.py 4 3 512 256 2>&1 | grep -v amdgpu | tail -3; } > $O/contention_$TAG.log 2>&1; cat $O/contention_$TAG.log | cut -c1-300;;
    sqattnf32) cd /tmp      # SQ / GRBM counters of the exact-fp32 attention kernel (k_attn<4>, csrc/vae_attn.hip) at T = 30 000
        (timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/sqa_$TAG -o p -- python $R/probes/attn_probe.py 30000 --exact-only 2>&1 | grep "^T=") > $O/sq_attnf32_$TAG.log 2>&1
        (timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sqb_$TAG -o p -- python $R/probes/attn_probe.py 30000 --exact-only 2>&1 | grep "^T=") >> $O/sq_attnf32_$TAG.log 2>&1
        cd $R; python tools/pmc_sq.py $O/pmc_sq_attnf32_$TAG.json $O/sqa_$TAG $O/sqb_$TAG 2>&1 | grep "k_attn\|kernel" | head -6 | tee -a $O/sq_attnf32_$TAG.log; rm -rf $O/sqa_$TAG $O/sqb_$TAG;;
    convf32) for f in "" 0 1 3; do MDTILE_CONVF32_FORM=$f timeout 600 python probes/convf32_probe.py 2>&1 | grep "^form"; done | tee $O/convf32_probe_$TAG.log;;
    chk:*) sel=${w#chk:}; (timeout 1800 python -m pytest ${sel//+/ } -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | grep -v "Tiled VAE\|amdgpu.ids\|Sampling" | tail -50) > $O/pytest_chk_$TAG.log 2>&1; tail -40 $O/pytest_chk_$TAG.log;;
    *) bash $R/tools/gpu_r5.sh $TAG $w;;
  esac
done
find gpurun_out -name "*.db" -delete 2>/dev/null
