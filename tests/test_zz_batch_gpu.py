"""B > 1 through the reference's decoder API (decoder.py:158-350 takes [B, nimgs, N, C]).  Batch elements are independent
in the reference (pinned by tests/test_oracle_vs_reference.py::test_batch_elements_are_independent_in_reference), so the
oracle for a batch is the B = 1 oracle per element; the HIP route must in addition be bit-identical to its own B = 1 calls.
(File named to run last: it is the newest test of the round.)"""
import pytest
import torch

from must3r_amd import synthetic as S
from must3r_amd.config import TINY
from util import TOL, rel_inf
from test_model_gpu import build

pytestmark = pytest.mark.gpu


def test_batched_decode_equals_per_scene_and_oracle():
    from oracle import must3r_ref as R
    cfg = TINY
    enc, dec = build(cfg, "fp16w2")
    sde, sdd = S.make_encoder_state_dict(cfg, 0), S.make_decoder_state_dict(cfg, 0)
    imgs, ts = S.make_images(6, 48, 64, 9)
    x, pos = enc(imgs.cuda(), ts.cuda())
    x, pos, t = x.view(2, 3, *x.shape[1:]), pos.view(2, 3, *pos.shape[1:]), ts.cuda().view(2, 3, 2)
    c = lambda *a: [v.contiguous() for v in a]
    mem, pm_a, feats = dec(*c(x[:, :2], pos[:, :2], t[:, :2]), None, return_feats=True)
    mem, pm_b = dec(*c(x[:, 2:], pos[:, 2:], t[:, 2:]), mem)
    mem_r, pm_r = dec(x, pos, t, mem, render=True)
    assert pm_a.shape == (2, 2, 48, 64, 7) and pm_b.shape == (2, 1, 48, 64, 7) and pm_r.shape == (2, 3, 48, 64, 7)
    assert mem[0][0].shape[:2] == (2, 36) and mem[1].shape == (2, 36) and tuple(mem[2:]) == (3, 3, 36)
    assert len(feats) == cfg.dec_depth + 1 and feats[0].shape[:3] == (2, 2, 12) and feats[-1].shape == (2, 2, 12, cfg.dec_dim)
    assert mem_r is mem
    # list form
    mem_l, pm_l = dec.forward_list([x[:, :2].contiguous()], [pos[:, :2].contiguous()], [t[:, :2].contiguous()], None)
    assert torch.equal(pm_l[0], pm_a)

    xo, po = R.encoder_forward(sde, cfg, imgs, ts)
    xo, po, to = xo.view(2, 3, *xo.shape[1:]), po.view(2, 3, *po.shape[1:]), ts.view(2, 3, 2)
    for b in range(2):
        s = slice(b, b + 1)
        m1, p1a, f1 = dec(*c(x[s, :2], pos[s, :2], t[s, :2]), None, return_feats=True)
        m1, p1b = dec(*c(x[s, 2:], pos[s, 2:], t[s, 2:]), m1)
        _, p1r = dec(*c(x[s], pos[s], t[s]), m1, render=True)
        assert torch.equal(pm_a[s], p1a) and torch.equal(pm_b[s], p1b) and torch.equal(pm_r[s], p1r)
        assert all(torch.equal(a[s], v) for a, v in zip(mem[0], m1[0])) and torch.equal(mem[1][s], m1[1])
        assert all(torch.equal(a[s], v) for a, v in zip(feats, f1))
        mo, oa = R.decoder_forward(sdd, cfg, *c(xo[s, :2], po[s, :2], to[s, :2]), None, False, "kv")
        mo, ob = R.decoder_forward(sdd, cfg, *c(xo[s, 2:], po[s, 2:], to[s, 2:]), mo, False, "kv")
        _, orr = R.decoder_forward(sdd, cfg, *c(xo[s], po[s], to[s]), mo, True, "kv")
        for got, want in ((pm_a[s], oa), (pm_b[s], ob), (pm_r[s], orr)):
            assert rel_inf(got.cpu(), torch.as_tensor(want)) < TOL["fp16w2"]
