"""GPU: the retrieval front-end through the C ABI (must3r_hip_affine, _row_norm, _topk_gather,
_weighted_spoc) and the drop-in classes of must3r_amd.retrieval, against the oracle (oracle/retrieval_ref.py) and the
reference-generated fixture.  Tolerances (floating point, relative to the largest magnitude of the output): whitening in
float64 like the reference -> 1e-6 (fp32 rounding of the result); projector in fp32 with exact MFMA products -> 2e-6;
pooled descriptors 5e-6; top-k indices must be identical (the test data has no attention ties)."""
import numpy as np
import pytest
import torch

from oracle import retrieval_ref as RR
from must3r_amd import synthetic as S
from util import load_golden, rel_inf
from test_ops_gpu import record

pytestmark = pytest.mark.gpu


class _Backbone(torch.nn.Module):
    def __init__(self, d):
        super().__init__()
        self.embed_dim = d


def _model(dim, nfeat, sd, hdims=None, **kw):
    from must3r_amd.retrieval import RetrievalModel
    m = RetrievalModel(_Backbone(dim), hdims=[dim] if hdims is None else hdims, nfeat=nfeat, **kw).cuda().eval()
    msg = m.load_state_dict(sd, strict=True)
    assert not msg.missing_keys and not msg.unexpected_keys
    return m


def test_retrieval_reference_fixture():
    g = load_golden("retrieval_small")
    x = torch.randn((3, 48, 256), generator=torch.Generator().manual_seed(5))
    for tag, kw in (("full", dict(prewhiten=-1, postwhiten=-1)), ("resid", dict(prewhiten=None, postwhiten=-1, residual=True))):
        sd = S.make_retrieval_state_dict(256, seed=3, prewhiten=kw["prewhiten"] is not None)
        m = _model(256, 20, sd, **kw)
        f, a, i = m.forward_local(x.cuda())
        gl = m.forward_global(x.cuda())
        assert np.array_equal(i.cpu().numpy(), g[tag + "/idx"])
        errs = dict(attn=rel_inf(a.cpu(), g[tag + "/attn"]), feat=rel_inf(f.cpu(), g[tag + "/feat"]), glob=rel_inf(gl.cpu(), g[tag + "/glob"]))
        record("retrieval_fixture", tag=tag, **errs)
        assert errs["attn"] < 2e-6 and errs["feat"] < 2e-6 and errs["glob"] < 5e-6, errs


@pytest.mark.parametrize("shape", [(1, 768, 1024, 300), (4, 672, 1024, 300), (2, 320, 1024, -0.5), (3, 100, 64, 1000)])
def test_retrieval_vs_oracle(shape):
    """the encoder geometry (768 tokens x 1024), mixed-resolution token counts, fractional nfeat, nfeat > N."""
    Bn, N, Cd, nfeat = shape
    sd = S.make_retrieval_state_dict(Cd, seed=Bn)
    x = torch.randn((Bn, N, Cd), generator=torch.Generator().manual_seed(N)) * (1.0 + torch.arange(N).view(1, N, 1) / N)
    m = _model(Cd, nfeat, sd, prewhiten=-1, postwhiten=-1)
    f, a, i = m.forward_local(x.cuda())
    gl = m.forward_global(x.cuda())
    fo, ao, io = RR.forward_local(sd, x, nfeat)
    go = RR.forward_global(sd, x)
    assert f.shape == fo.shape and i.dtype == torch.int64
    same = float((i.cpu() == io).float().mean())
    errs = dict(attn=rel_inf(a.cpu(), ao), glob=rel_inf(gl.cpu(), go), idx_same=same)
    if same == 1.0:
        errs["feat"] = rel_inf(f.cpu(), fo)
    record("retrieval_vs_oracle", shape=list(shape), **errs)
    assert same == 1.0, same
    assert errs["attn"] < 2e-6 and errs["feat"] < 2e-6 and errs["glob"] < 5e-6, errs
    assert bool((a[:, :-1] >= a[:, 1:]).all())      # sorted descending like torch.topk


def test_multilayer_projector_and_l2norm_whitener():
    """build_projector with hidden layers (retrieval/model.py:139-151: Linear - LayerNorm - GELU stacks) and Whitener(l2norm=dim) (:77-78)
    against the reference-generated fixture and, at the encoder geometry, against the oracle.  fp32 LayerNorm / erf GELU: 3e-6."""
    from must3r_amd.retrieval import Whitener
    g = load_golden("retrieval_small")
    sd = S.make_retrieval_state_dict(256, seed=4, hdims=[320, 192])
    x = torch.randn((3, 48, 256), generator=torch.Generator().manual_seed(6))
    m = _model(256, 20, sd, hdims=[320, 192], prewhiten=-1, postwhiten=-1)
    assert [k for k in m.state_dict()] == list(sd.keys()) or set(m.state_dict()) == set(sd)
    f, a, i = m.forward_local(x.cuda())
    gl = m.forward_global(x.cuda())
    assert np.array_equal(i.cpu().numpy(), g["deep/idx"])
    errs = dict(attn=rel_inf(a.cpu(), g["deep/attn"]), feat=rel_inf(f.cpu(), g["deep/feat"]), glob=rel_inf(gl.cpu(), g["deep/glob"]))
    for dim in (-1, 1):
        wh = Whitener(256, l2norm=dim).cuda()
        wh.load_state_dict({"m": sd["prewhiten.m"], "p": sd["prewhiten.p"]})
        errs[f"l2norm{dim}"] = rel_inf(wh(x.cuda()).cpu(), g[f"l2norm{dim}/out"])
    # the encoder geometry: 1024 -> 2048 -> 512 -> 1024 with a residual
    sd2 = S.make_retrieval_state_dict(1024, seed=9, hdims=[2048, 512, 1024])
    x2 = torch.randn((2, 768, 1024), generator=torch.Generator().manual_seed(8)) * (1.0 + torch.arange(768).view(1, 768, 1) / 768)
    m2 = _model(1024, 300, sd2, hdims=[2048, 512, 1024], prewhiten=-1, postwhiten=-1, residual=True)
    f2, a2, i2 = m2.forward_local(x2.cuda())
    fo, ao, io = RR.forward_local(sd2, x2, 300, True)
    same = float((i2.cpu() == io).float().mean())
    errs.update(attn_1024=rel_inf(a2.cpu(), ao), glob_1024=rel_inf(m2.forward_global(x2.cuda()).cpu(), RR.forward_global(sd2, x2, True)), idx_same=same)
    if same == 1.0:
        errs["feat_1024"] = rel_inf(f2.cpu(), fo)
    wh0 = Whitener(1024, l2norm=0).cuda()        # a leading dimension: the strided kernel with inner = N * C
    wh0.load_state_dict({"m": sd2["prewhiten.m"], "p": sd2["prewhiten.p"]})
    errs["l2norm0"] = rel_inf(wh0(x2.cuda()).cpu(), RR.whiten(x2, sd2["prewhiten.m"], sd2["prewhiten.p"], l2norm=0))
    record("retrieval_multilayer", **errs)
    assert same == 1.0 and all(v < 3e-6 for k, v in errs.items() if k != "idx_same"), errs


def test_whitener_is_float64_inside_and_affine_options():
    """the Whitener must not lose the float64 centre/projection (large mean, tiny spread), bias / residual / transposed B."""
    from must3r_amd.retrieval import affine
    g = torch.Generator().manual_seed(2)
    x = 1000.0 + torch.randn((300, 96), generator=g) * 1e-3
    m = torch.full((1, 96), 1000.0, dtype=torch.float64)
    p = torch.randn((96, 96), generator=g, dtype=torch.float64)
    ref = ((x.double() - m) @ p).float()
    out = affine(x.cuda(), m, p, b_transposed=False, double=True).cpu()
    assert rel_inf(out, ref) < 1e-6
    w, b = torch.randn((70, 96), generator=g), torch.randn((70,), generator=g)
    r = torch.randn((300, 70), generator=g)
    ref2 = (x.double() @ w.double().t() + b.double() + r.double()).float()
    out2 = affine(x.cuda(), None, w, b_transposed=True, bias=b, resid=r.cuda()).cpu()
    assert rel_inf(out2, ref2) < 2e-6


def test_topk_ties_and_errors():
    from must3r_amd.retrieval import how_select_local, weighted_spoc
    feat = torch.arange(2 * 6 * 8, dtype=torch.float32).view(2, 6, 8).cuda()
    attn = torch.tensor([[1.0, 3.0, 3.0, 0.5, 3.0, 2.0], [0.0, 0.0, 0.0, 0.0, 0.0, 0.0]]).cuda()
    f, a, i = how_select_local(feat, attn, 4)
    assert i.cpu().tolist() == [[1, 2, 4, 5], [0, 1, 2, 3]]              # ties: lower index first
    assert torch.equal(f[0, 0], feat[0, 1]) and a.cpu().tolist()[0] == [3.0, 3.0, 3.0, 2.0]
    z = weighted_spoc(feat, attn)
    assert torch.isfinite(z).all() and float(z[1].abs().max()) == 0.0      # zero weights -> zero vector (F.normalize eps)
    with pytest.raises(RuntimeError):
        how_select_local(feat.cpu(), attn.cpu(), 2)                        # CPU tensors: no fallback
