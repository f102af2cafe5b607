"""CPU checks of the VAE oracle (oracle/vae_ref.py; parity unpinned: the network is third-party, see its header) and of the synthetic weights'
key list: the restatement is self-consistent, and the things the native graph relies on hold in the definition itself."""
import torch
import torch.nn.functional as F

from oracle import vae_ref
from univst_amd import synth

CFG = dict(vae_ref.SVD_VAE_CONFIG, block_out_channels=(32, 32, 64, 64))


def _sd(seed=3):
    return {k: v.float() for k, v in synth.vae_state_dict(CFG, device="cpu", dtype=torch.float32, seed=seed).items()}


def test_frame_conv_is_conv3d_311():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 6, 5, 4, 3, generator=g)
    w = torch.randn(7, 6, 3, 1, 1, generator=g)
    b = torch.randn(7, generator=g)
    assert torch.allclose(vae_ref.frame_conv(x, w, b), F.conv3d(x, w, b, padding=(1, 0, 0)), atol=1e-5)


def test_matmul_convolution_path_equals_conv2d():
    """vae_ref.CONV_VIA_MATMUL (what the 16 x 512 x 512 device oracle of tests/test_gpu_vae.py runs on): im2col + matmul against F.conv2d — 3x3 pad 1,
    3x3 stride 2 behind the asymmetric pad, 1x1, with and without bias, chunked over images."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(4)
    try:
        vae_ref.CONV_VIA_MATMUL = True
        for (n, cin, cout, H, W, k, pad, stride, bias) in [(3, 8, 16, 10, 12, 3, 1, 1, True), (2, 16, 8, 9, 9, 3, 0, 2, True), (5, 8, 8, 6, 7, 1, 0, 1, False)]:
            x, w = torch.randn(n, cin, H, W, generator=g), torch.randn(cout, cin, k, k, generator=g)
            b = torch.randn(cout, generator=g) if bias else None
            got = vae_ref._conv2d(x, w, b, padding=pad, stride=stride)
            want = F.conv2d(x, w, b, padding=pad, stride=stride)
            assert got.shape == want.shape and (got - want).abs().max() < 1e-4
    finally:
        vae_ref.CONV_VIA_MATMUL = False


def test_key_list_and_shapes_of_the_svd_vae():
    sd = synth.vae_state_dict(vae_ref.SVD_VAE_CONFIG, device="cpu", dtype=torch.float16)
    # counts of the published class at layers_per_block = 2: 12 decoder up resnets + 2 mid, 8 encoder down resnets + 2 mid
    assert sum(k.endswith("time_mixer.mix_factor") for k in sd) == 14
    assert sd["decoder.up_blocks.2.resnets.0.spatial_res_block.conv_shortcut.weight"].shape == (256, 512, 1, 1)
    assert sd["decoder.up_blocks.3.resnets.0.temporal_res_block.conv1.weight"].shape == (128, 128, 3, 1, 1)
    assert sd["decoder.time_conv_out.weight"].shape == (3, 3, 3, 1, 1) and sd["quant_conv.weight"].shape == (8, 8, 1, 1)
    assert sd["encoder.down_blocks.1.resnets.0.conv_shortcut.weight"].shape == (256, 128, 1, 1)
    assert "encoder.down_blocks.3.downsamplers.0.conv.weight" not in sd and "decoder.up_blocks.3.upsamplers.0.conv.weight" not in sd
    assert sd["decoder.mid_block.attentions.0.to_q.weight"].shape == (512, 512)
    n = sum(v.numel() for v in sd.values())
    assert 95e6 < n < 100e6, n            # the published checkpoint has 97.7 M parameters


def test_decode_shapes_clip_independence_and_temporal_coupling():
    sd = _sd()
    z = torch.randn(8, 4, 4, 4, generator=torch.Generator().manual_seed(1))
    y = vae_ref.decode(sd, z, 4, CFG)
    assert y.shape == (8, 3, 32, 32) and torch.isfinite(y).all()
    # two clips in one batch == each clip alone (5-D GroupNorm and Conv3d see one clip at a time) ...
    assert torch.allclose(y[:4], vae_ref.decode(sd, z[:4], 4, CFG), atol=1e-5)
    # ... and frames of a clip are coupled
    assert (y[:1] - vae_ref.decode(sd, z[:1], 1, CFG)).abs().max() > 1e-3


def test_encode_moments_shape_and_asymmetric_downsample_padding():
    sd = _sd(5)
    x = torch.rand(2, 3, 32, 32, generator=torch.Generator().manual_seed(2)) * 2 - 1
    m = vae_ref.encode_moments(sd, x, CFG)
    assert m.shape == (2, 8, 4, 4)
    # Downsample2D(padding=0) pads the bottom / right only: the top-left output pixel does not see an image shifted in from the right
    x2 = x.clone()
    x2[..., :, -1] += 1.0
    m2 = vae_ref.encode_moments(sd, x2, CFG)
    assert (m2 - m).abs().max() > 1e-4
