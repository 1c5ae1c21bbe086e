# Functional runs of the N-rank bench flow on ONE device (all ranks on cuda:0 over gloo), with the per-tile error of rank 0's assembled image.
cd ${GRAFT_REPO_ROOT:-$PWD}
run() { name=$1; shift
  out=$( (env "$@" 2>/dev/null) | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); c = d['debug_check']
    print(d['n_gpus'], d['ms_per_step'], c['assembled_image_rel_err_vs_single_rank'], [(t['tile'], t['owner'], float('%.1e' % t['rel_err'])) for t in c.get('per_tile', [])])
except Exception as e: print('ERR', e)")
  echo "$name: $out"; }
B="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass --evals 2"
for spec in "$@"; do
  case $spec in
    a) run "N=8 L=512 t256 SP=0"       MDTILE_SP_ESTIMATOR=0 $B --gpus 8 --latent 512 --vae-tile 256;;
    b) run "N=2 L=1024 t256"           $B --gpus 2 --latent 1024 --vae-tile 256;;
    c) run "N=2 L=1024 t256 TB=1"      MDTILE_TILE_BATCH=1 $B --gpus 2 --latent 1024 --vae-tile 256;;
    d) run "N=2 L=1024 t256 LW=0"      MDTILE_LIVE_WINDOW=0 $B --gpus 2 --latent 1024 --vae-tile 256;;
    e) run "N=2 L=1024 t256 SP=0"      MDTILE_SP_ESTIMATOR=0 $B --gpus 2 --latent 1024 --vae-tile 256;;
    f) run "N=8 L=512 t256 SP=0 TB=1"  MDTILE_SP_ESTIMATOR=0 MDTILE_TILE_BATCH=1 $B --gpus 8 --latent 512 --vae-tile 256;;
    g) run "N=8 L=1024 t256"           $B --gpus 8 --latent 1024 --vae-tile 256;;
    h) run "N=2 L=512 t256 SP=0"       MDTILE_SP_ESTIMATOR=0 $B --gpus 2 --latent 512 --vae-tile 256;;
  esac
done
