"""TEST INFRASTRUCTURE -- CPU restatement of the retrieval front-end.

Only ``tests/`` may import this file.  Functional (state-dict in, tensors out) restatement of
must3r/retrieval/model.py: ``Whitener.forward`` (:67-79), ``weighted_spoc`` (:82-88), ``how_select_local`` (:91-101),
``RetrievalModel.extract_features_and_attention`` / ``forward_local`` / ``forward_global`` (:165-180), the projector
(``build_projector`` :139-151: Linear, or Linear - LayerNorm - GELU stacks ending in a Linear).  PINNED: the reference module imports verbatim (oracle/ref_shims.py stubs only its
unused image loader); ``tests/test_oracle_vs_reference.py::test_retrieval_equals_reference`` compares, and
``tests/golden/retrieval_small.npz`` holds reference outputs.
"""
import torch


def whiten(x, m, p, l2norm=None):
    """Whitener.forward (:67-79): float64 centre + projection (+ F.normalize along ``l2norm``), cast back to the input dtype."""
    shape = x.shape
    y = torch.matmul(x.reshape(-1, shape[-1]).to(torch.float64) - m.to(torch.float64), p.to(torch.float64)).view(shape)
    if l2norm is not None:
        y = torch.nn.functional.normalize(y, dim=l2norm)
    return y.to(x.dtype)


def project(sd, pre):
    """nn.Sequential of build_projector (:139-151) read from its state-dict keys: ``projector.{3j}`` Linear, ``projector.{3j+1}``
    LayerNorm (eps 1e-5, the nn.LayerNorm default), GELU (erf) between hidden layers; the last Linear has no norm / activation."""
    idx = sorted({int(k.split(".")[1]) for k in sd if k.startswith("projector.")})
    lin = [i for i in idx if sd[f"projector.{i}.weight"].dim() == 2]
    h = pre
    for j, i in enumerate(lin):
        h = torch.nn.functional.linear(h, sd[f"projector.{i}.weight"], sd[f"projector.{i}.bias"])
        if j + 1 < len(lin):
            h = torch.nn.functional.layer_norm(h, (h.shape[-1],), sd[f"projector.{i + 1}.weight"], sd[f"projector.{i + 1}.bias"], 1e-5)
            h = torch.nn.functional.gelu(h)
    return h


def extract_features_and_attention(sd, x, residual=False):
    pre = whiten(x, sd["prewhiten.m"], sd["prewhiten.p"]) if "prewhiten.m" in sd else x
    if "projector.0.weight" in sd:
        proj = project(sd, pre) + (pre if residual else 0.0)
    else:
        proj = pre + (pre if residual else 0.0)
    attention = proj.norm(dim=-1)
    post = whiten(proj, sd["postwhiten.m"], sd["postwhiten.p"]) if "postwhiten.m" in sd else proj
    return post, attention


def weighted_spoc(feat, attn):
    return torch.nn.functional.normalize((feat * attn[:, :, None]).sum(dim=1), dim=1)


def how_select_local(feat, attn, nfeat):
    nfeat = int(-nfeat * feat.size(1)) if nfeat < 0 else int(nfeat)
    topk_attn, topk_indices = torch.topk(attn, min(nfeat, attn.size(1)), dim=1)
    return torch.gather(feat, 1, topk_indices.unsqueeze(-1).expand(-1, -1, feat.size(2))), topk_attn, topk_indices


def forward_local(sd, x, nfeat=300, residual=False):
    feat, attn = extract_features_and_attention(sd, x, residual)
    return how_select_local(feat, attn, nfeat)


def forward_global(sd, x, residual=False):
    feat, attn = extract_features_and_attention(sd, x, residual)
    return weighted_spoc(feat, attn)
