"""Determinism / batch-invariance diagnostics on the GPU."""
import sys, math, torch, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from must3r_amd import _lib as lib
from test_ops_gpu import run_gemm, P, stream
torch.manual_seed(0)
tdt = torch.float16
A = torch.randn((3072, 1024), device="cuda").to(tdt)
W = (torch.randn((1024, 1024), device="cuda") / 32).to(tdt)
b = torch.randn((1024,), device="cuda")
o1 = torch.empty((3072, 1024), device="cuda", dtype=tdt); o2 = torch.empty((768, 1024), device="cuda", dtype=tdt)
o3 = torch.empty_like(o1)
run_gemm(lib, "fp16", lib.EPI_STORE16, A, W, b, o1)
run_gemm(lib, "fp16", lib.EPI_STORE16, A[:768], W, b, o2)
run_gemm(lib, "fp16", lib.EPI_STORE16, A, W, b, o3)
print("gemm 128-tile rerun equal:", torch.equal(o1, o3), " 128 vs 64 tile equal:", torch.equal(o1[:768], o2), (o1[:768].float()-o2.float()).abs().max().item())
f1 = torch.empty((3072, 1024), device="cuda"); f2 = torch.empty((768, 1024), device="cuda")
run_gemm(lib, "fp16", lib.EPI_F32, A, W, b, f1); run_gemm(lib, "fp16", lib.EPI_F32, A[:768], W, b, f2)
print("gemm f32 128 vs 64:", torch.equal(f1[:768], f2), (f1[:768]-f2).abs().max().item())
# attention: 5 views vs one
heads, N = 4, 768
D = heads*64
qkv = (torch.randn((5*N, 3*D), device="cuda")*1.5).to(tdt)
def attn(views, q, k, v, Rq):
    o = torch.zeros((Rq, D), device="cuda", dtype=tdt)
    tab = torch.tensor(views, dtype=torch.int32, device="cuda")
    lib.check(lib.load().must3r_hip_op_attention(1, P(q), P(k), P(v), P(o), q.stride(0), k.stride(0), v.stride(0), o.stride(0), heads, P(tab), len(views), N, 0, None, 0, stream()))
    torch.cuda.synchronize(); return o
q, k, v = qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:]
oa = attn([(i*N, N, i*N, N, 0, 0) for i in range(5)], q, k, v, 5*N)
ob = attn([(i*N, N, i*N, N, 0, 0) for i in range(5)], q, k, v, 5*N)
oc = attn([(3*N, N, 3*N, N, 0, 0)], q, k, v, 5*N)
print("attn rerun equal:", torch.equal(oa, ob), " batched vs single view equal:", torch.equal(oa[3*N:4*N], oc[3*N:4*N]), (oa[3*N:4*N].float()-oc[3*N:4*N].float()).abs().max().item())
# encoder SMALL
from must3r_amd import synthetic as S
from must3r_amd.config import SMALL, MUST3R_512
import must3r_amd.model as M
for cfg, H, Wd in ((SMALL, 224, 224),):
    enc = M.Dust3rEncoder(img_size=(cfg.img_size,)*2, embed_dim=cfg.enc_dim, depth=cfg.enc_depth, num_heads=cfg.enc_heads)
    enc.load_state_dict(S.make_encoder_state_dict(cfg, 0)); enc = enc.cuda().eval()
    imgs, ts = S.make_images(5, H, Wd, 0); imgs, ts = imgs.cuda(), ts.cuda()
    xa, _ = enc(imgs, ts); xb, _ = enc(imgs, ts); xc, _ = enc(imgs[3:4], ts[3:4])
    print(cfg.enc_dim, "enc rerun equal:", torch.equal(xa, xb), " batched vs single:", torch.equal(xa[3], xc[0]), (xa[3]-xc[0]).abs().max().item())
    # layer-by-layer: depth 1..3
    for d in range(1, cfg.enc_depth+1):
        e2 = M.Dust3rEncoder(img_size=(cfg.img_size,)*2, embed_dim=cfg.enc_dim, depth=d, num_heads=cfg.enc_heads)
        sd = {k_: v_ for k_, v_ in S.make_encoder_state_dict(cfg, 0).items() if not k_.startswith("blocks_enc.") or int(k_.split(".")[1]) < d}
        e2.load_state_dict(sd); e2 = e2.cuda().eval()
        ya, _ = e2(imgs, ts); yc, _ = e2(imgs[3:4], ts[3:4])
        print("  depth", d, "batched vs single max diff", (ya[3]-yc[0]).abs().max().item())
