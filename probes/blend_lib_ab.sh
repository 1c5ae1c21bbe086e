# Same-box A/B of the blend kernel between two builds of libmdtile.so (8K grid; cold = 8 rotating buffer sets, warm = static buffers):
#   bash probes/blend_lib_ab.sh probes/_ab/<other>.so
OTHER=$1
L=multidiffusion-upscaler-for-automatic1111_amd/mdtile/libmdtile.so
cp $L /tmp/libmdtile_current.so
for r in 1 2; do
  for which in other current; do
    if [ $which = other ]; then cp $OTHER $L; else cp /tmp/libmdtile_current.so $L; fi
    (timeout 300 python bench.py --no-vae --steps 20 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); b = d.get('roofline_blend') or d.get('roofline')
print('$which', 'cold us', b.get('avg_us'), 'warm us', (b.get('warm') or {}).get('avg_us'))")
    (timeout 120 python - <<'PY'
import sys, torch, hashlib
sys.path.insert(0, "multidiffusion-upscaler-for-automatic1111_amd"); sys.path.insert(0, ".")
import mdtile as E
dev = torch.device("cuda:0"); torch.manual_seed(0)
W = H = 1024; tw = th = 128; N, C = 2, 4
plan = E.Plan(W, H, tw, th, 8, 8)
packed = torch.randn(plan.num_tiles * N, C, th, tw, device=dev)
weights = torch.zeros(H, W, device=dev); E.weight_map_add_grid(plan, None, weights)
out = torch.empty(N, C, H, W, device=dev)
E.BlendCall(plan, E.METHOD_MD, [packed], N, C, weights=weights, out=out, packed=True)()
torch.cuda.synchronize()
print("   output sha1", hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16])
PY
    )
  done
done
cp /tmp/libmdtile_current.so $L
