"""Retrieval front-end on the encoder tokens -- drop-in for the model part of must3r/retrieval/model.py (:59-183).

``Whitener`` and ``RetrievalModel`` keep the reference's constructor arguments, attribute names and state-dict keys
(``prewhiten.m/p`` and ``postwhiten.m/p`` in float64, ``projector.{i}.weight/bias``), so ``load_state_dict`` of a
reference retrieval checkpoint works unchanged; ``forward_local`` / ``forward_global`` take the encoder tokens
``x [B,N,C]`` (cuda fp32) like the reference's (demo/inference.py:40).  Every stage is a native call: float64 centre +
PCA projection (``must3r_hip_affine``, fp64 MFMA; ``Whitener(l2norm=dim)`` -> ``must3r_hip_l2_normalize``), the projector Linears (fp32
MFMA, exact products; hidden layers of a multi-layer projector: ``must3r_hip_layernorm_act_f32``), token attention
(``must3r_hip_row_norm``), top-k selection + gather (``must3r_hip_topk_gather``) or weighted sum pooling
(``must3r_hip_weighted_spoc``).  No CPU fallback.  Learning the whitening (``pcawhitenlearn_shrinkage``) and the ASMK
codebook stay on the host as in the reference (out of scope: SURVEY.md section 8f rank 4 is the front-end only).
"""
import torch
import torch.nn as nn

from . import _lib


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _tokens(x, what):
    if not isinstance(x, torch.Tensor) or not x.is_cuda:
        raise RuntimeError(f"must3r_amd.retrieval: {what} must be a CUDA tensor; the HIP path has no CPU fallback")
    return x.float().contiguous()


def affine(x, sub, B, b_transposed, bias=None, resid=None, double=False):
    """out = (x - sub) @ B (+ bias) (+ resid) over the last dimension, fp32 in / out; float64 inside when ``double``."""
    x = _tokens(x, "x")
    K = x.shape[-1]
    N = B.shape[0] if b_transposed else B.shape[1]
    M = x.numel() // K
    td = torch.float64 if double else torch.float32
    dev = x.device
    subd = None if sub is None else sub.detach().to(device=dev, dtype=td).reshape(-1).contiguous()
    Bd = B.detach().to(device=dev, dtype=td).contiguous()
    biasd = None if bias is None else bias.detach().to(device=dev, dtype=td).contiguous()
    residd = None if resid is None else _tokens(resid, "resid")
    out = torch.empty((*x.shape[:-1], N), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().must3r_hip_affine(1 if double else 0, x.data_ptr(), None if subd is None else subd.data_ptr(), Bd.data_ptr(),
                                                 1 if b_transposed else 0, None if biasd is None else biasd.data_ptr(),
                                                 None if residd is None else residd.data_ptr(), out.data_ptr(), M, N, K, _stream(x)))
    return out


class Whitener(nn.Module):
    """retrieval/model.py:59-79."""

    def __init__(self, dim, l2norm=None):
        super().__init__()
        self.m = nn.Parameter(torch.zeros((1, dim)).double())
        self.p = nn.Parameter(torch.eye(dim, dim).double())
        self.l2norm = l2norm

    @torch.no_grad()
    def forward(self, x):
        out = affine(x, self.m, self.p, b_transposed=False, double=True).view(x.shape)
        if self.l2norm is not None:
            out = l2_normalize(out, self.l2norm)   # the reference normalises the float64 product (:77-78); here its fp32 rounding
        return out.to(x.dtype)


def l2_normalize(x, dim):
    """F.normalize(x, dim=dim) (eps 1e-12), in place on a contiguous fp32 tensor."""
    x = _tokens(x, "x")
    dim = dim % x.dim()
    outer = 1
    for d in x.shape[:dim]:
        outer *= d
    inner = 1
    for d in x.shape[dim + 1:]:
        inner *= d
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().must3r_hip_l2_normalize(x.data_ptr(), outer, x.shape[dim], inner, x.data_ptr(), _stream(x)))
    return x


def layernorm_act(x, ln, gelu):
    """nn.LayerNorm ``ln`` (+ erf GELU) on fp32 rows."""
    x = _tokens(x, "x")
    Cd = x.shape[-1]
    if tuple(ln.normalized_shape) != (Cd,):
        raise ValueError(f"LayerNorm over {tuple(ln.normalized_shape)} on rows of {Cd}")
    dev = x.device
    w = None if ln.weight is None else ln.weight.detach().to(device=dev, dtype=torch.float32).contiguous()
    b = None if ln.bias is None else ln.bias.detach().to(device=dev, dtype=torch.float32).contiguous()
    out = torch.empty_like(x)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().must3r_hip_layernorm_act_f32(x.data_ptr(), None if w is None else w.data_ptr(), None if b is None else b.data_ptr(),
                                                            float(ln.eps), x.numel() // Cd, Cd, 1 if gelu else 0, out.data_ptr(), _stream(x)))
    return out


def weighted_spoc(feat, attn):
    """retrieval/model.py:82-88."""
    feat, attn = _tokens(feat, "feat"), _tokens(attn, "attn")
    Bn, N, Cd = feat.shape
    out = torch.empty((Bn, Cd), dtype=torch.float32, device=feat.device)
    with torch.cuda.device(feat.device):
        _lib.check(_lib.load().must3r_hip_weighted_spoc(feat.data_ptr(), attn.data_ptr(), Bn, N, Cd, out.data_ptr(), _stream(feat)))
    return out


def how_select_local(feat, attn, nfeat):
    """retrieval/model.py:91-101 -> (topk_features [B,k,C], topk_attn [B,k], topk_indices int64 [B,k])."""
    feat, attn = _tokens(feat, "feat"), _tokens(attn, "attn")
    Bn, N, Cd = feat.shape
    if nfeat < 0:
        assert nfeat >= -1.0
        nfeat = int(-nfeat * N)
    else:
        nfeat = int(nfeat)
    k = min(nfeat, N)
    of = torch.empty((Bn, k, Cd), dtype=torch.float32, device=feat.device)
    oa = torch.empty((Bn, k), dtype=torch.float32, device=feat.device)
    oi = torch.empty((Bn, k), dtype=torch.int64, device=feat.device)
    with torch.cuda.device(feat.device):
        _lib.check(_lib.load().must3r_hip_topk_gather(feat.data_ptr(), attn.data_ptr(), Bn, N, Cd, k, of.data_ptr(), oa.data_ptr(), oi.data_ptr(),
                                                      _stream(feat)))
    return of, oa, oi


class RetrievalModel(nn.Module):
    """retrieval/model.py:104-183 (inference part).  ``backbone`` is only used for its ``embed_dim`` (the reference's
    forward paths take encoder tokens, not images)."""

    def __init__(self, backbone, freeze_backbone=1, prewhiten=None, hdims=[1024], residual=False, postwhiten=None, featweights="l2norm",
                 nfeat=300, pretrained_retrieval=None):
        super().__init__()
        self.freeze_backbone = freeze_backbone
        try:
            self.backbone_dim = backbone.enc_embed_dim
        except Exception:
            self.backbone_dim = backbone.embed_dim
        self.prewhiten = nn.Identity() if prewhiten is None else Whitener(self.backbone_dim)
        self.prewhiten_freq = prewhiten
        self.residual = residual
        if residual:
            assert hdims[-1] == self.backbone_dim
        self.projector = self.build_projector(hdims, residual)
        self.dim = hdims[-1] if len(hdims) > 0 else self.backbone_dim
        self.postwhiten_freq = postwhiten
        self.postwhiten = nn.Identity() if postwhiten is None else Whitener(self.dim)
        if featweights != "l2norm":
            raise NotImplementedError(featweights)
        self.featweights = featweights
        self.nfeat = nfeat
        for prm in self.parameters():
            prm.requires_grad = False
        if pretrained_retrieval is not None:
            ckpt = torch.load(pretrained_retrieval, "cpu", weights_only=False)
            msg = self.load_state_dict(ckpt["model"], strict=False)
            assert len(msg.unexpected_keys) == 0 and all(k.startswith("backbone") or k.startswith("postwhiten") for k in msg.missing_keys)

    def build_projector(self, hdims, residual):   # retrieval/model.py:139-151: same modules, same state-dict keys
        d = self.backbone_dim
        if len(hdims) == 0:
            return nn.Identity()
        layers = []
        for h in hdims[:-1]:
            layers += [nn.Linear(d, h), nn.LayerNorm(h), nn.GELU()]
            d = h
        layers.append(nn.Linear(d, hdims[-1]))
        return nn.Sequential(*layers)

    def _project(self, pre):
        """projector(pre) (+ pre): every Linear one fp32 GEMM, LayerNorm + GELU of a hidden layer one kernel."""
        mods = list(self.projector)
        h = pre
        for i in range(0, len(mods) - 1, 3):
            lin, ln, act = mods[i], mods[i + 1], mods[i + 2]
            assert isinstance(lin, nn.Linear) and isinstance(ln, nn.LayerNorm) and isinstance(act, nn.GELU)
            h = layernorm_act(affine(h, None, lin.weight, b_transposed=True, bias=lin.bias), ln, gelu=True)
        lin = mods[-1]
        return affine(h, None, lin.weight, b_transposed=True, bias=lin.bias, resid=pre if self.residual else None)

    @torch.no_grad()
    def extract_features_and_attention(self, x):   # retrieval/model.py:165-172
        x = _tokens(x, "x")
        pre = x if isinstance(self.prewhiten, nn.Identity) else self.prewhiten(x)
        if isinstance(self.projector, nn.Identity):
            proj = pre if not self.residual else pre + pre
        else:
            proj = self._project(pre)
        Bn, N, Cd = proj.shape
        attention = torch.empty((Bn, N), dtype=torch.float32, device=proj.device)
        with torch.cuda.device(proj.device):
            _lib.check(_lib.load().must3r_hip_row_norm(proj.data_ptr(), Bn * N, Cd, attention.data_ptr(), _stream(proj)))
        post = proj if isinstance(self.postwhiten, nn.Identity) else self.postwhiten(proj)
        return post, attention

    def forward_local(self, x):
        feat, attn = self.extract_features_and_attention(x)
        return how_select_local(feat, attn, self.nfeat)

    def forward_global(self, x):
        feat, attn = self.extract_features_and_attention(x)
        return weighted_spoc(feat, attn)

    def forward(self, x):
        return self.forward_global(x)
