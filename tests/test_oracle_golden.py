"""CPU: the oracle (oracle/*.py) against the golden vectors produced by the reference's own modules
(tests/golden/make_golden.py).  fp32; tolerance 1e-4 relative to max|ref| (observed <= 1.1e-6)."""
import numpy as np
import pytest
import torch

from oracle import unet_ref, pipeline_ref, maskprop_ref, flow_ref, synth_inputs as si

TOL = 1e-4


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-12)).item()


def test_adains(golden):
    g = golden("g1_attention_adain")
    assert rel(unet_ref.attention_adain(g["cnt"], g["sty"]), g["out"]) < TOL
    g = golden("g1_latent_adain")
    assert rel(unet_ref.latent_adain(g["cnt"], g["sty"]), g["out"]) < TOL


def test_sparse_causal_and_pnp_closure_components(golden):
    """G3 / G2: the reference's SparseCausalAttention (three index modes) and its PnP closure patched onto it
    (pnp_utils.py:20-100, idx 0 / 13 / 25 inside the window, 26 outside) vs the oracle with the same parameters."""
    g = golden("g3_sparse_causal")
    x, sd = g["x"], g["params"]
    for tag, index in (("stock", [-1, 0, "first"]), ("pnp", [-1, "first"]), ("first_only", ["first"])):
        q, k, v = (unet_ref._lin(sd, "m." + n, x, bias=False) for n in ("to_q", "to_k", "to_v"))
        y = unet_ref._lin(sd, "m.to_out.0", unet_ref.sdpa(q, unet_ref.sparse_causal_gather(k, 4, index),
                                                           unet_ref.sparse_causal_gather(v, 4, index), 8))
        assert rel(y, g["out"][tag]) < TOL
    assert rel(unet_ref.attn1_forward(sd, "m", x, 4, 8, None), g["out"]["stock"]) < TOL
    g2 = golden("g2_pnp_closure")
    for idx in (0, 13, 25, 26):
        assert rel(unet_ref.attn1_forward(sd, "m", x, 4, 8, dict(idx=idx)), g2["out"][f"idx{idx}"]) < TOL


def test_unet_building_blocks(golden):
    """G4: PseudoConv3d, ResnetBlockPseudo3D (5-D GroupNorm), Up/DownsamplePseudo3D, SpatioTemporalTransformerBlock / Model
    of the reference at C = 32 with NON-trivial temporal parameters, vs the oracle functions."""
    g = golden("g4_components")
    x5, temb, xt, ctx, out, prm = g["x5"], g["temb"], g["xt"], g["ctx"], g["out"], g["params"]
    assert rel(unet_ref.pseudo_conv3d(prm["conv"], "m", x5, exact_temporal=True), out["conv"]) < TOL
    assert rel(unet_ref.resnet_block(prm["resnet"], "m", x5, temb, 8, 1e-5, exact_temporal=True), out["resnet"]) < TOL
    assert rel(unet_ref.upsample(prm["up"], "m", x5, exact_temporal=True), out["up"]) < TOL
    dp = "m.op" if "m.op.weight" in prm["down"] else "m.conv"
    assert rel(unet_ref.pseudo_conv3d(prm["down"], dp, x5, stride=2, padding=1, exact_temporal=True), out["down"]) < TOL
    assert rel(unet_ref.transformer_block(prm["block"], "m", xt, ctx, 3, 4, None, exact_temporal=True), out["block"]) < TOL
    assert rel(unet_ref.transformer_model(prm["model"], "m", x5, ctx[:2], 4, 8, None, exact_temporal=True), out["model"]) < TOL


@pytest.mark.parametrize("tag,trivial", [("trivial", True), ("general", False)])
def test_tiny_unet(golden, tag, trivial):
    cfg = unet_ref.TINY_CONFIG
    sd = unet_ref.synth_state_dict(cfg, seed=33, trivial_temporal=trivial)
    F_, h_, w_ = 4, 16, 16
    x = torch.cat([si.content_latent(50, F_, h_, w_), si.style_latent(50, F_, h_, w_),
                   si.content_latent(49, F_, h_, w_)])
    ctx = si.text_embedding(cfg["cross_attention_dim"]).expand(3, -1, -1).contiguous()
    g = golden(f"g5_{tag}_single")
    eps, feats = unet_ref.unet_forward(sd, cfg, x[:1], 301, ctx[:1], None, ft_indices=[2])
    assert rel(eps, g["eps"]) < TOL
    if g["feat"] is not None:
        assert rel(feats[2], g["feat"].float()) < 2e-3      # stored as fp16
    for idx in (0, 25, 26):
        g = golden(f"g5_{tag}_pnp{idx}")
        eps, _ = unet_ref.unet_forward(sd, cfg, x, g["t"], ctx, pnp_idx=idx, exact_temporal=not trivial)
        assert rel(eps, g["eps"]) < TOL


def test_ddim(golden):
    g = golden("g7_ddim")
    s = pipeline_ref.DDIMSchedule()
    s.set_timesteps(50)
    assert torch.equal(s.timesteps, g["timesteps"])
    for k, t in enumerate(s.timesteps):
        assert rel(s.next_step(g["e"], t, g["z"]), g["next"][k]) < 1e-6
        assert rel(s.step(g["e"], t, g["z"])[0], g["prev"][k]) < 1e-6


def test_maskprop_bit_exact(golden):
    g = golden("g8_maskprop")
    torch.manual_seed(33)
    masks = np.stack(maskprop_ref.video_mask_propagation(si.maskprop_features(), si.soft_first_mask()))
    ref = np.unpackbits(g["masks"].numpy(), axis=-1).astype(bool)
    assert np.array_equal(masks[0], g["frame0"].numpy())
    assert np.array_equal(masks[1:] > 0, ref[1:])
    assert set(np.unique(masks[1:])) <= {0, 255}


def test_flow_pieces(golden):
    g = golden("g9_flow")
    H = W = 64
    fwd = si.translation_flow(H, W, 4.0, -2.0, 1)
    bwd = si.translation_flow(H, W, -4.0, 2.0, 2)
    bwd[10:20, 10:30] += 3.0
    occ = flow_ref.compute_occlusion_mask(fwd, bwd, threshold=1.5)
    assert np.array_equal(occ, g["occ"].numpy())
    rs = np.random.RandomState(5)
    img, orig = rs.randint(0, 256, (H, W, 3)).astype(np.uint8), rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
    assert np.array_equal(flow_ref.apply_mask(img, occ, orig), g["applied"].numpy())


def test_remap_identity_and_shift():
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, (32, 48, 3)).astype(np.uint8)
    gx, gy = np.meshgrid(np.arange(48), np.arange(32))
    assert np.array_equal(flow_ref.remap_bilinear_u8(img, gx.astype(np.float32), gy.astype(np.float32)), img)
    out = flow_ref.remap_bilinear_u8(img, gx.astype(np.float32) + 1, gy.astype(np.float32))
    assert np.array_equal(out[:, :-1], img[:, 1:]) and (out[:, -1] == 0).all()
    half = flow_ref.remap_bilinear_u8(img, gx.astype(np.float32) + 0.5, gy.astype(np.float32))
    exp = (img[:, :-1].astype(np.int64) * 16384 + img[:, 1:].astype(np.int64) * 16384 + 16384) >> 15
    assert np.array_equal(half[:, :-1], exp.astype(np.uint8))


@pytest.mark.parametrize("tag", ["nomask", "mask"])
def test_transfer_loop(golden, tag):
    """the reference's own 50-step three-branch loop (tiny UNet, F=16) vs the oracle loop; every branch of
    the step logic is sampled: i in {0,25,26,40,41,45,46,49}."""
    g = golden(f"g10_{tag}")
    cfg = unet_ref.TINY_CONFIG
    sd = unet_ref.synth_state_dict(cfg, seed=33)
    F_, h_, w_ = 16, 16, 16
    ctx = si.text_embedding(cfg["cross_attention_dim"]).expand(3, -1, -1).contiguous()
    ci = [si.content_latent(k, F_, h_, w_) for k in range(51)]
    sy = [si.style_latent(k, F_, h_, w_) for k in range(51)]
    m01 = None
    if tag == "mask":
        m01 = torch.from_numpy(pipeline_ref.mask_from_png_values(si.disc_masks(F_, h_ * 8, w_ * 8)))[None]
    got = {}
    keep = (0, 25, 26, 40, 41, 45, 46, 49)
    with torch.no_grad():
        pipeline_ref.video_style_transfer_loop(
            lambda x, t, i: unet_ref.unet_forward(sd, cfg, x, int(t), ctx, pnp_idx=i, exact_temporal=False)[0],
            pipeline_ref.DDIMSchedule(), unet_ref.latent_adain(ci[50], sy[50]), ci, sy, m01, 50,
            callback=lambda i, t, l: got.__setitem__(i, l.clone()) if i in keep else None)
    for i in keep:
        assert rel(got[i], g[f"i{i}"]) < 2e-3, i


def test_sd2_shaped_unet(golden):
    cfg = unet_ref.TINY_SD2_CONFIG
    sd = unet_ref.synth_state_dict(cfg, seed=33)
    x = torch.cat([si.content_latent(50, 4, 16, 16), si.style_latent(50, 4, 16, 16), si.content_latent(49, 4, 16, 16)])
    ctx = si.text_embedding(cfg["cross_attention_dim"]).expand(3, -1, -1).contiguous()
    g = golden("g11_sd2")
    assert rel(unet_ref.unet_forward(sd, cfg, x[:1], 301, ctx[:1], None)[0], g["single"]) < TOL
    assert rel(unet_ref.unet_forward(sd, cfg, x, 781, ctx, pnp_idx=10)[0], g["pnp10"]) < TOL


@pytest.mark.parametrize("tag,easy", [("ddim_loop", False), ("ddim_loop_plus", True)])
def test_inversion_loops(golden, tag, easy):
    """G12: the reference's own ddim_loop / ddim_loop_plus (inversion_tools/ddim_inversion.py:87-167; Easy-Inv averaging
    for i = 3..12 applied after eps) and the t=301 feature dump of its UNet forward vs the oracle loop."""
    g = golden("g12_inversion")[tag]
    cfg = unet_ref.TINY_CONFIG
    sd = unet_ref.synth_state_dict(cfg, seed=33)
    text = si.text_embedding(cfg["cross_attention_dim"])
    z0 = 0.7 * si.content_latent(0, 4, 16, 16)
    osch = pipeline_ref.DDIMSchedule()
    osch.set_timesteps(50)
    dump = {}

    def eps_fn(z, t, i):
        e, f = unet_ref.unet_forward(sd, cfg, z, int(t), text, None, ft_indices=[2] if int(t) == 301 else None, exact_temporal=False)
        if f:
            dump["feat"] = f[2]
        return e
    with torch.no_grad():
        mine = pipeline_ref.ddim_inversion_loop(eps_fn, osch, z0, 50, easy)
    for k in (1, 3, 4, 12, 13, 14, 50):
        assert rel(mine[k], g[f"k{k}"]) < TOL, k
    assert rel(dump["feat"], g["feat"]) < TOL


def test_get_warp_composition(golden):
    """G13: the reference's get_warp (cal_optica_flow.py:51-99) with RAFT replaced by seeded flows and cv2.remap by the
    restated fixed-point remap: which flow warps, the 1.5 px occlusion threshold, the composite over ref_image1."""
    H = W = 96
    f_fwd = si.translation_flow(H, W, 3.3, -2.7, 101, noise=0.6)
    f_bwd = si.translation_flow(H, W, -3.3, 2.7, 102, noise=0.6)
    f_bwd[5:15, 20:40] += 4.0
    rs = np.random.RandomState(13)
    im1, im2 = rs.randint(0, 256, (H, W, 3)).astype(np.uint8), rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
    k = [0]

    def flow_fn(a, b):
        k[0] += 1
        return (f_fwd, f_bwd)[(k[0] - 1) % 2]
    assert np.array_equal(flow_ref.get_warp(flow_fn, im1, im2), golden("g13_get_warp")["warped"].numpy())


def test_pixel_smoother_leg_of_the_loop():
    """the smoothing leg (stable_diffusion.py:713-759) inside the oracle loop: active exactly for i in [20,25), changes the
    trajectory from step 20 on, and asks for 2 flows per warp x 58 boundary-clipped warps per step."""
    cfg = unet_ref.TINY_CONFIG
    sd = unet_ref.synth_state_dict(cfg, seed=33)
    F_, h_, w_ = 16, 8, 8
    ctx = si.text_embedding(cfg["cross_attention_dim"]).expand(3, -1, -1).contiguous()
    ci = [si.content_latent(k, F_, h_, w_) for k in range(51)]
    sy = [si.style_latent(k, F_, h_, w_) for k in range(51)]
    m01 = torch.from_numpy(pipeline_ref.mask_from_png_values(si.disc_masks(F_, h_ * 8, w_ * 8)))[None]
    vae = si.FakeLinearVAE()
    z = torch.randn(3, 4, h_, w_)
    assert (vae.encode_tensor(vae.decode_tensor(z)) - z).abs().max() < 1e-4
    calls = []

    def run(with_smoother):
        osch = pipeline_ref.DDIMSchedule()
        osch.set_timesteps(50)
        flow = si.CountingFlow(h_ * 8, w_ * 8)
        sm = None
        if with_smoother:
            inner = pipeline_ref.pixel_smoother(osch, vae.decode_tensor, vae.encode_tensor, flow, m01.numpy())

            def sm(i, t, lat, eps):
                calls.append(i)
                return inner(i, t, lat, eps)
        got = {}
        with torch.no_grad():
            pipeline_ref.video_style_transfer_loop(
                lambda x, t, i: unet_ref.unet_forward(sd, cfg, x, int(t), ctx, pnp_idx=i, exact_temporal=False)[0],
                osch, unet_ref.latent_adain(ci[50], sy[50]), ci, sy, m01, 50, smoother=sm,
                callback=lambda i, t, l: got.__setitem__(i, l.clone()))
        return got, flow.k
    a, _ = run(False)
    b, nflow = run(True)
    assert calls == [20, 21, 22, 23, 24]
    assert nflow == 5 * 2 * 58                      # 16 key frames x up to 4 neighbours, boundary-clipped = 58 warps per step
    assert all(torch.equal(a[i], b[i]) for i in range(20)) and not torch.allclose(a[20], b[20], atol=1e-3)


@pytest.mark.parametrize("tag", ["gs1", "gs7p5"])
def test_reconstruction_loop(golden, tag):
    """G14: the reference's reconstruction loop (stable_diffusion.py:478-628) = plain DDIM sampling with the single-branch UNet;
    with guidance_scale 7.5 and the empty negative prompt of the scripts both halves of the CFG batch see the same text, so the
    guided noise equals the plain one (uncond + w * (text - uncond))."""
    g = golden("g14_reconstruction")[tag]
    cfg = unet_ref.TINY_CONFIG
    sd = unet_ref.synth_state_dict(cfg, seed=33)
    text = si.text_embedding(cfg["cross_attention_dim"])
    osch = pipeline_ref.DDIMSchedule()
    osch.set_timesteps(50)
    z = si.content_latent(50, 16, 16, 16)
    with torch.no_grad():
        for i, t in enumerate(osch.timesteps):
            e = unet_ref.unet_forward(sd, cfg, z, int(t), text, None, exact_temporal=False)[0]
            z, _ = osch.step(e, t, z)
            if i in (0, 10, 25, 49):
                assert rel(z, g[f"i{i}"]) < 2e-3, i


def test_product_scheduler_step_matches_g7(golden):
    """univst_amd.schedulers.DDIMScheduler.step (the API-completeness method; the loops use the folded axpby kernel) and its
    timestep table against G7 (restated diffusers 0.35.1 DDIM, eta = 0) over all 50 timesteps."""
    from univst_amd.schedulers import DDIMScheduler
    g = golden("g7_ddim")
    s = DDIMScheduler()
    s.set_timesteps(50)
    assert torch.equal(s.timesteps, g["timesteps"])
    for i, t in enumerate(g["timesteps"]):
        out = s.step(g["e"], int(t), g["z"])
        assert torch.equal(out.prev_sample, g["prev"][i]), int(t)


# ---- SURVEY §8f-4 groundwork: the reference-owned pieces of the SD3 / SD3.5 path (oracle/sd3_ref.py; no HIP path exists yet)
def test_sd3_adains(golden):
    from oracle import sd3_ref
    a = golden("g15_sd3_attention_adain")
    assert rel(sd3_ref.attention_adain(a["cnt"], a["sty"]), a["out"]) < TOL
    b = golden("g15_sd3_latent_adain")
    assert rel(sd3_ref.latent_adain(b["cnt"], b["sty"]), b["out"]) < TOL


def test_sd3_joint_attention_processors(golden):
    """CrossFrameProcessor and AttentionShiftProcessor of the reference's SD3 plugin (the latter under the documented fixed reading
    thresh2 == eta2) on 3 branches x 16 frames, 9 image + 5 text tokens, inside and outside the shift window."""
    from oracle import sd3_ref
    g = golden("g16_sd3_processors")
    P, hid, enc = g["params"], g["hidden"], g["enc"]
    img, txt = sd3_ref.joint_attention(P, 2, hid, enc)
    assert rel(img, g["cross_frame"]["img"]) < TOL and rel(txt, g["cross_frame"]["txt"]) < TOL
    assert rel(sd3_ref.joint_attention(P, 2, hid, None), g["cross_frame_no_text"]) < TOL
    for idx in (0, 17, 30, 31):
        img, txt = sd3_ref.joint_attention(P, 2, hid, enc, idx=idx, shift=True, eta1=0.0, eta2=0.6)
        assert rel(img, g[f"shift_idx{idx}"]["img"]) < TOL and rel(txt, g[f"shift_idx{idx}"]["txt"]) < TOL, idx
    # the first two branches never change; the third does inside the window only
    img30, _ = sd3_ref.joint_attention(P, 2, hid, enc, idx=30, shift=True)
    plain, _ = sd3_ref.joint_attention(P, 2, hid, enc)
    assert torch.equal(img30[:32], plain[:32]) and not torch.equal(img30[32:], plain[32:])
    # cross-frame gather: frame f reads ['first', f-1, f] of its own clip
    x = torch.arange(48.0).view(48, 1, 1, 1).expand(48, 1, 2, 1)
    got = sd3_ref.cross_frame_gather(x)[:, 0, ::2, 0]
    assert got[0].tolist() == [0, 0, 0] and got[5].tolist() == [0, 4, 5] and got[16].tolist() == [16, 16, 16] and got[47].tolist() == [32, 46, 47]


def test_sd3_rectified_flow_inversions(golden):
    from oracle import sd3_ref
    g = golden("g17_sd3_rf")
    sig, z0 = g["sigmas"], g["z0"]
    assert torch.equal(sig, sd3_ref.flow_match_sigmas(10)) and sig[-1] == 0 and bool((sig[:-1] > sig[1:]).all())

    def vel(x, t1000, idx):
        tt = (t1000 / 1000.0).reshape(-1)[0]
        return torch.tanh(0.7 * x.flip(-1)) * (0.5 + tt) - 0.3 * x + 0.05 * idx
    assert rel(sd3_ref.rf_inversion(vel, z0.clone(), sig, g["rf_inversion"]["noise"], 0.5)[-1], g["rf_inversion"]["final"]) < TOL
    assert rel(sd3_ref.rf_solver(vel, z0.clone(), sig)[-1], g["rf_solver"]["final"]) < TOL


def test_sd3_processors_head_dim_64_g18(golden):
    """G18: the reference's processors at a shape the native kernels serve (head_dim 64, 40 image + 13 text tokens) and its
    attention_adain on [16, 2, 40, 64]; also the (parity-unpinned) JointTransformerBlock restatement runs on such shapes."""
    from oracle import sd3_ref
    g = golden("g18_sd3_processors_hd64")
    P, hid, enc = g["params"], g["hidden"], g["enc"]
    img, txt = sd3_ref.joint_attention(P, 2, hid, enc)
    assert rel(img, g["cross_frame"]["img"]) < TOL and rel(txt, g["cross_frame"]["txt"]) < TOL
    assert rel(sd3_ref.joint_attention(P, 2, hid, None), g["cross_frame_no_text"]) < TOL
    for idx in (0, 17, 30, 31):
        img, txt = sd3_ref.joint_attention(P, 2, hid, enc, idx=idx, shift=True, eta1=0.0, eta2=0.6)
        assert rel(img, g[f"shift_idx{idx}"]["img"]) < TOL and rel(txt, g[f"shift_idx{idx}"]["txt"]) < TOL, idx
    a = g["attention_adain"]
    assert rel(sd3_ref.attention_adain(a["cnt"], a["sty"]), a["out"]) < TOL


def test_sd3_pipeline_loops_g19(golden):
    """G19: the reference's own CustomStableDiffusion3Pipeline.video_style_transfer (no mask) and .reconstruction over the closed-form
    velocity field; eta tables; and the product's FlowMatchEuler tables equal the oracle's restatement (third-party, unpinned)."""
    from oracle import sd3_ref
    from univst_amd.schedulers import FlowMatchEulerDiscreteScheduler
    g = golden("g19_sd3_pipeline_loops")
    ti = sd3_ref.toy_loop_inputs()
    Fr = ti["content"][0].shape[0]
    ts, sig = sd3_ref.flow_match_schedule(50)
    sch = FlowMatchEulerDiscreteScheduler()
    sch.set_timesteps(50)
    assert torch.equal(sch.timesteps, ts) and torch.equal(sch.sigmas, sig) and sig[-1] == 0 and abs(float(ts[-1]) - 8.9286) < 1e-3
    eta = sd3_ref.generate_eta_values(ts, 25, 39, 0.85, "constant")
    assert eta[24] == 0 and eta[25] == 0.85 and eta[38] == 0.85 and eta[39] == 0
    for trend in ("linear_increase", "linear_decrease"):
        assert rel(torch.tensor([float(v) for v in sd3_ref.generate_eta_values(ts, 10, 20, 0.95, trend)]), g[f"eta_{trend}"]) < TOL
    vf = lambda x, t, i: sd3_ref.toy_velocity(x, t, i, Fr)          # noqa: E731
    start = sd3_ref.latent_adain(ti["content"][50], ti["style"][50])
    out = sd3_ref.sd3_transfer_loop(vf, start, ti["content"][0], ti["content"], ti["style"], ts, sig, eta)
    assert rel(out, g["video_style_transfer"]) < TOL
    rec = sd3_ref.sd3_reconstruction_loop(vf, ti["content"][0], ti["content"][50], ts, sig, eta)
    assert rel(rec, g["reconstruction"]) < TOL
    # the mask (fixed reading of the undefined name) changes the result, and only through the masked pixels' history
    outm = sd3_ref.sd3_transfer_loop(vf, start, ti["content"][0], ti["content"], ti["style"], ts, sig, eta, mask=ti["mask"])
    assert not torch.allclose(outm, out)
