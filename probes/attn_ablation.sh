# Ablations of k_attn_bf16x3<512> at T = 77 284 (WRONG numbers, same MFMA stream): which part of a key block's 38.4 k cycles is what.
#   variants are builds of csrc/vae_attn_bf16x3.hip with -DMDT_ATTN_ABLATE=<bits> (profiles/r5g/attn_ablation_variants.diff):
#   1 = score phase without its LDS fragment reads, 2 = without the K / Q slab DMA, 4 = output phase without P (LDS) and V^T (global) loads
#   bash probes/attn_ablation.sh        (needs probes/_ab/libmdtile_attn_abl_<v>.so)
L=multidiffusion-upscaler-for-automatic1111_amd/mdtile/libmdtile.so
cp $L /tmp/libmdtile_current.so
for v in 0 1 2 3 4 7 0; do
  if [ $v = 0 ]; then cp /tmp/libmdtile_current.so $L; else cp probes/_ab/libmdtile_attn_abl_$v.so $L; fi
  echo "ablate=$v $(timeout 200 python probes/attn_probe.py 77284 --quick 2>&1 | grep 'T=')"
done
cp /tmp/libmdtile_current.so $L
