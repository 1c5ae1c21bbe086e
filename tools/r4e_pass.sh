mkdir -p gpurun_out; O=gpurun_out
(timeout 300 probes/cu_mempipe_probe 2>&1) > $O/r4e_cu_mempipe.log; cat $O/r4e_cu_mempipe.log
(timeout 600 python -m pytest tests/test_gpu_rec.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -5) > $O/r4e_pytest_rec.log 2>&1; cat $O/r4e_pytest_rec.log
q="--steps 1 --warmup 1 --no-cpu-baseline --no-f32-pass --no-whole-tile-pass --no-oracle-pass"
for b in 1 auto; do
  if [ $b = 1 ]; then export MDTILE_REC_BLOCKS=1; else unset MDTILE_REC_BLOCKS; fi
  for cfg in "--latent 1024 --vae-tile 256" "--latent 512 --vae-tile 64" "--latent 256 --vae-tile 64" "--latent 512 --vae-tile 128"; do
    timeout 600 python bench.py $q $cfg 2>&1 | tail -1 > $O/r4e_tmp.json
    python -c "
import json; d=json.load(open('$O/r4e_tmp.json')); r=d['roofline']; print('blocks=$b', '$cfg', d['ms_per_step'], r['kernel'], r['frac'], {k: v for k, v in r['breakdown_s'].items() if 'rec' in k})" | tee -a $O/r4e_bench_heuristic_ab.log
  done
done
