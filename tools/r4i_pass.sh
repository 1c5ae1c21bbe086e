mkdir -p gpurun_out; O=$PWD/gpurun_out; R=$PWD; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_shard.py tests/test_gpu_vae.py tests/test_gpu_rec.py -m gpu -q --tb=short -p no:cacheprovider -k "bench_flow or stacked or env_alone or live_windows or two_blocks" 2>&1 | tail -15) > $O/r4i_pytest.log 2>&1; tail -8 $O/r4i_pytest.log
(timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32-pass 2>&1 | tail -1) > $O/r4i_bench.json
python -c "
import json; d=json.load(open('$O/r4i_bench.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['stage_ms'], d['roofline_blend']['avg_us'], d['roofline_blend']['frac'], d['roofline_blend']['warm']); print(d['roofline']['breakdown_s']); print(d['whole_tiles']); print({k: v for k, v in d['parity'].items() if 'err' in k and 'tiles' not in k})"
cd /tmp
(timeout 300 rocprofv3 --att --att-target-cu 1 --kernel-include-regex "k_attn_bf16x3" -d $O/r4i_att -o att -- python $R/probes/attn_probe.py 8000 2>&1 | tail -25) > $O/r4i_att.log 2>&1; tail -12 $O/r4i_att.log; ls -la $O/r4i_att 2>/dev/null | head; du -sh $O/r4i_att 2>/dev/null
