# Same-box A/B of two builds of libmdtile.so on the 8K bench step:  bash tools/lib_ab.sh probes/_ab/<other>.so [rounds]
# (the other library is built by hand from another commit's csrc/ with mdtile/build.py's flags; probes/_ab/ is git-ignored)
OTHER=$1; ROUNDS=${2:-2}
L=multidiffusion-upscaler-for-automatic1111_amd/mdtile/libmdtile.so
cp $L /tmp/libmdtile_current.so
for r in $(seq $ROUNDS); do
  for which in other current; do
    if [ $which = other ]; then cp $OTHER $L; else cp /tmp/libmdtile_current.so $L; fi
    (timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32-pass --no-oracle-pass --no-whole-tile-pass --no-profile-pass 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$which', d['ms_per_step'])")
  done
done
cp /tmp/libmdtile_current.so $L
