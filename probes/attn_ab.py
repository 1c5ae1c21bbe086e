"""In-process A/B of two libmdtile builds' split-bf16 attention (run on the GPU box): the single-tile attention time is bimodal ACROSS
processes on this pool (profiles/r3g), so two kernels can only be compared inside one process, alternating.
usage: python probes/attn_ab.py <other_lib.so> [T ...]      (A = the in-tree libmdtile.so, B = other_lib)"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A = ctypes.CDLL(os.path.join(ROOT, "multidiffusion-upscaler-for-automatic1111_amd", "mdtile", "libmdtile.so"))
B = ctypes.CDLL(sys.argv[1])
for L in (A, B):
    L.mdtile_vae_attn_ws_size.restype = ctypes.c_size_t
    L.mdtile_vae_attn_ws_size.argtypes = [ctypes.c_int] * 3
    L.mdtile_vae_attn.restype = ctypes.c_int
    L.mdtile_vae_attn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
C = 512
Ts = [int(a) for a in sys.argv[2:]] or [30000, 77284]
for T in Ts:
    torch.manual_seed(0)
    q, k, v = torch.randn(1, C, T, device=dev), torch.randn(1, C, T, device=dev) * 1.5, torch.randn(1, T, C, device=dev)
    ws = torch.empty(max(A.mdtile_vae_attn_ws_size(1, C, T), B.mdtile_vae_attn_ws_size(1, C, T)), dtype=torch.uint8, device=dev)
    outs = {}
    def run(L, o):
        rc = L.mdtile_vae_attn(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), 1, C, T, C ** -0.5, 0, ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
    oa, ob = torch.empty_like(q), torch.empty_like(q)
    run(A, oa), run(B, ob)
    torch.cuda.synchronize()
    print(f"T={T}: A vs B max dev {((oa - ob).abs().max() / ob.abs().max()).item():.2e}", flush=True)
    flops = 4.0 * T * T * C
    for rep in range(4):
        line = f"  rep {rep}: "
        for name, L, o in (("A(tree)", A, oa), ("B(other)", B, ob)):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                run(L, o)
            e.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(e) / 3
            line += f"{name} {ms:8.3f} ms {flops / ms * 1e-9:6.1f} TF   "
        print(line, flush=True)
