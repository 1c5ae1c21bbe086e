# Round-6 GPU pass.   usage: bash tools/gpu_r6.sh <tag> [what...]     (through gpurun; artefacts land in gpurun_out/)
# what: every step of tools/gpu_r5.sh / gpu_r3.sh (tests | smoke | bench | benchq | prof | pmc | sqbench | encode | slow | ...) plus
#   r6new    this round's new GPU tests (bare / single-process bench forms, encoder stress parity, statistics calls, stream copy)
#   bare8    `python bench.py --gpus 8 --steps 1 --warmup 0` with NO launcher environment on this one-GPU box (rc, the JSON line)
#   sp8      `python bench.py --gpus 8 --single-process --steps 1 --warmup 1` (cuda:0 listed eight times)
#   blendab  probes/blend_r6_ab.py: the shipping blend kernel, the LDS-staged form and the stream-copy floor, cold and warm, same process
#   attnab   probes/attn_ab.py <other .so>: two builds of the attention kernel alternated in one process
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
    blendab) (timeout 600 python probes/blend_r6_ab.py 2>&1 | grep -v amdgpu.ids) > $O/blend_r6_ab_$TAG.log 2>&1; cat $O/blend_r6_ab_$TAG.log;;
    attnab) (timeout 900 python probes/attn_ab.py probes/_ab/libmdtile_attn_base.so 2>&1 | grep -v amdgpu.ids) > $O/attn_ab_$TAG.log 2>&1; cat $O/attn_ab_$TAG.log;;
    chk:*) sel=${w#chk:}; (timeout 1800 python -m pytest ${sel//+/ } -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | grep -v "Tiled VAE\|amdgpu.ids\|Sampling" | tail -50) > $O/pytest_chk_$TAG.log 2>&1; tail -40 $O/pytest_chk_$TAG.log;;
    *) bash $R/tools/gpu_r5.sh $TAG $w;;
  esac
done
find gpurun_out -name "*.db" -delete 2>/dev/null
