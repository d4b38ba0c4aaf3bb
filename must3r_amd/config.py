"""Model geometry of the multi-view forward path.

Mirrors the constructor arguments of the reference modules
(``Dust3rEncoder.__init__`` must3r/model/encoder.py:14-23, ``MUSt3R.__init__``
must3r/model/decoder.py:19-37).  The two released geometries are ``MUSt3R_224`` and ``MUSt3R_512``
(ViT-L/16 encoder, ViT-B decoder; README.md:107-131); smaller geometries exist only so the parity
tests can run the CPU oracle in seconds.
"""
from dataclasses import dataclass, asdict

HEAD_DIM = 64  # 1024/16 (encoder) and 768/12 (decoder); the HIP attention kernels are built for it


@dataclass(frozen=True)
class ModelConfig:
    img_size: int = 224          # square training size S; actual inputs are S x {S, 3S/4, ...}
    patch_size: int = 16
    enc_dim: int = 1024
    enc_depth: int = 24
    enc_heads: int = 16
    dec_dim: int = 768
    dec_depth: int = 12
    dec_heads: int = 12
    mlp_ratio: int = 4
    rope_freq: float = 100.0     # 'RoPE100' (pos_embed.py:20)
    rope_f0: float = 1.0         # F0 = old/new size when run off the native size (pos_embed.py:12-19)

    @property
    def output_dim(self):        # 16*16*7 = 1792 (decoder.py:24)
        return self.patch_size * self.patch_size * 7

    def validate(self):
        assert self.enc_dim == self.enc_heads * HEAD_DIM, "encoder head dim must be 64"
        assert self.dec_dim == self.dec_heads * HEAD_DIM, "decoder head dim must be 64"
        assert self.patch_size == 16
        for d in (self.enc_dim, self.dec_dim):
            assert d % 128 == 0, "feature dims must be multiples of 128 (GEMM tile)"
        return self

    def to_dict(self):
        return asdict(self)


MUST3R_224 = ModelConfig(img_size=224)
MUST3R_512 = ModelConfig(img_size=512)
# test-only geometries (same code paths, CPU-oracle friendly)
TINY = ModelConfig(img_size=64, enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=128, dec_depth=2, dec_heads=2)
SMALL = ModelConfig(img_size=224, enc_dim=256, enc_depth=3, enc_heads=4, dec_dim=128, dec_depth=3, dec_heads=2)
