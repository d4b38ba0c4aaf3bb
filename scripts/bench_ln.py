"""LayerNorm microbench: the shapes of the S = 20 step, time, achieved bytes/s, output digest (run under M3R_LN_ROWS=0 / 1: same bits)."""
import ctypes as C, hashlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from must3r_amd import _lib as lib
L = lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
for name, M, Cc in (("enc chunk", 30720, 1024), ("dec update", 15360, 768), ("dec render", 307200, 768), ("one view", 768, 768)):
    x = torch.randn((M, Cc), device="cuda") * 3 + 0.5
    w = torch.randn((Cc,), device="cuda"); b = torch.randn((Cc,), device="cuda")
    out = torch.empty((M, Cc), device="cuda", dtype=torch.float16)
    def run():
        lib.check(L.must3r_hip_op_layernorm(1, P(x), None, P(w), P(b), P(out), None, None, None, M, Cc, 1e-6, st))
    run(); torch.cuda.synchronize()
    sha = hashlib.sha1(out.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:10]
    ref = torch.nn.functional.layer_norm(x[:512].double(), (Cc,), w.double(), b.double(), 1e-6)
    err = float((out[:512].double() - ref).abs().max())
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"  {name:11s} M={M:6d} C={Cc:4d} {us:8.2f} us {M * Cc * 6 / us / 1e6:6.2f} TB/s  err {err:.1e}  sha {sha}")
print(f"M3R_LN_ROWS={os.environ.get('M3R_LN_ROWS', '(default)')}")
