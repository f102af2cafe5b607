"""Seeded synthetic inputs shared by the golden generator, the tests, smoke() and bench.py's cpu_baseline
leg.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
import numpy as np
import torch


def _randn(shape, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float32)


def content_latent(k: int, F: int, h: int, w: int) -> torch.Tensor:
    """ddim_latents_k of the content inversion: N(0,1), seed 1000+k."""
    return _randn((1, 4, F, h, w), 1000 + k)


def style_latent(k: int, F: int, h: int, w: int) -> torch.Tensor:
    """ddim_latents_k of the style inversion: one frame repeated F times (ddim_inversion.py:51-53) plus
    1e-3 N(0,1); seed 2000+k."""
    base = _randn((1, 4, 1, h, w), 2000 + k).expand(1, 4, F, h, w)
    return (base + 1e-3 * _randn((1, 4, F, h, w), 3000 + k)).contiguous()


def text_embedding(D: int) -> torch.Tensor:
    return _randn((1, 77, D), 7)


def disc_masks(F: int, H: int, W: int) -> np.ndarray:
    """F frames uint8 {0,255}: a disc of radius H/4 translating 4*(H/512) px per frame."""
    yy, xx = np.mgrid[0:H, 0:W]
    out = np.zeros((F, H, W), np.uint8)
    r = H / 4.0
    step = 4.0 * H / 512.0
    for f in range(F):
        cx = W / 2.0 - step * F / 2 + step * f
        cy = H / 2.0
        out[f][(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] = 255
    return out


def maskprop_features(F=16, h=16, w=16, C=64, seed=11) -> torch.Tensor:
    """Spatially coherent synthetic UNet features [F,h,w,C] (fp16 like the reference's dump): smooth
    background field + an object signature moving with the disc + small per-frame noise."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    low = torch.randn(1, C, 4, 4, generator=g)
    bg = torch.nn.functional.interpolate(low, size=(h, w), mode="bilinear", align_corners=False)[0]   # [C,h,w]
    sig = torch.randn(C, generator=g)
    m = torch.from_numpy(disc_masks(F, h * 8, w * 8)[:, 4::8, 4::8].astype(np.float32) / 255.0)       # [F,h,w]
    feats = []
    for f in range(F):
        x = bg + 1.5 * sig[:, None, None] * m[f][None] + 0.15 * torch.randn(C, h, w, generator=g)
        feats.append(x.permute(1, 2, 0))
    return torch.stack(feats).to(torch.float16)


def soft_first_mask(H=128, W=128) -> np.ndarray:
    """Anti-aliased 'L' first-frame mask (multi-valued like examples/masks/mallard-fly.png: SURVEY X5)."""
    yy, xx = np.mgrid[0:H, 0:W]
    step = 4.0 * H / 512.0
    cx = W / 2.0 - step * 16 / 2
    d = np.sqrt((xx - cx) ** 2 + (yy - H / 2.0) ** 2)
    return np.clip((H / 4.0 - d) * 64.0 + 128.0, 0, 255).astype(np.uint8)


def translation_flow(H: int, W: int, dx: float, dy: float, seed: int, noise=0.25) -> np.ndarray:
    g = np.random.RandomState(seed)
    f = np.empty((H, W, 2), np.float32)
    f[..., 0] = dx
    f[..., 1] = dy
    return f + (noise * g.randn(H, W, 2)).astype(np.float32)


class FakeLinearVAE(torch.nn.Module):
    """Deterministic stand-in for the SVD temporal VAE in smoothing-leg tests (no VAE weights exist on either box):
    decode = fixed 1x1 channel mix 4 -> 3*64 + pixel_shuffle(8), encode = pixel_unshuffle(8) + its pseudo-inverse, so
    encode(decode(z)) == z up to rounding.  Duck-types what the pipeline touches: ``.config.scaling_factor``,
    ``.config.block_out_channels`` (len 4 => x8), ``decode(z, num_frames=).sample``, ``encode(x).latent_dist.sample()``."""

    def __init__(self, seed=21):
        super().__init__()
        import types
        g = torch.Generator(device="cpu").manual_seed(seed)
        wd = 0.35 * torch.randn(192, 4, generator=g)
        self.wd = torch.nn.Parameter(wd, requires_grad=False)
        self.we = torch.nn.Parameter(torch.linalg.pinv(wd.double()).float(), requires_grad=False)
        self.config = types.SimpleNamespace(scaling_factor=0.18215, block_out_channels=(1, 1, 1, 1))

    def forward(self, x, num_frames=1):
        return x

    def decode_tensor(self, z):
        y = torch.nn.functional.conv2d(z, self.wd.to(z.dtype)[:, :, None, None])
        return torch.nn.functional.pixel_shuffle(y, 8)

    def encode_tensor(self, x):
        y = torch.nn.functional.pixel_unshuffle(x, 8)
        return torch.nn.functional.conv2d(y, self.we.to(x.dtype)[:, :, None, None])

    def decode(self, z, num_frames=1):
        import types
        return types.SimpleNamespace(sample=self.decode_tensor(z))

    def encode(self, x):
        import types
        z = self.encode_tensor(x)
        return types.SimpleNamespace(latent_dist=types.SimpleNamespace(sample=lambda: z, mode=lambda: z))


class CountingFlow:
    """RAFT stand-in for the smoothing-leg tests: the k-th call returns a seeded analytic translation field (+ noise, + a
    patch that trips the occlusion test) that depends only on k — native and oracle loops ask in the same order
    (key frame ascending, neighbour ascending, forward then backward: stable_diffusion.py:731-747, cal_optica_flow.py:78-79)."""

    def __init__(self, H, W):
        self.H, self.W, self.k = H, W, 0

    def __call__(self, a=None, b=None):
        k = self.k
        self.k += 1
        sgn = 1.0 if k % 2 == 0 else -1.0
        d = 1.0 + (k // 2) % 3
        f = translation_flow(self.H, self.W, sgn * 1.3 * d, -sgn * 0.7 * d, 500 + k, noise=0.2)
        if k % 2 == 1:
            f[self.H // 8:self.H // 4, self.W // 4:self.W // 2] += 3.0
        return f
