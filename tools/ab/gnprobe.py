import sys, os, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from univst_amd import synth
from univst_amd.backbones.video_diffusion_sd import pnp_utils
g = torch.Generator().manual_seed(23)
x = torch.randn(3, 4, 16, 64, 64, generator=g).half().cuda()
ctx = torch.randn(1, 77, 768, generator=g).half().cuda().expand(3, -1, -1).contiguous()
for pert in (0.0, 25.0):
    unet = synth.build_unet(device="cuda", seed=9)
    with torch.no_grad():
        for n, p_ in unet.named_parameters():
            if n.endswith("conv1.bias") and pert:
                p_.add_(pert * torch.sign(torch.randn(p_.shape, generator=g)).to(p_))
    pipe = types.SimpleNamespace(unet=unet)
    pnp_utils.register_spatial_attention_pnp(pipe)
    pnp_utils.register_time(pipe, 12)
    outs = {}
    for name, opts in (("sep", {"gn_producer": 0}), ("prod", {"gn_producer": 1}), ("sep_nofold", {"gn_producer": 0, "ln_fold": 0})):
        for k, v in {"gn_producer": 1, "ln_fold": 1, **opts}.items():
            unet.set_native_option(k, v)
        outs[name] = unet(x, 741, encoder_hidden_states=ctx).sample.float()
    def d(a, b):
        return ((a - b).abs().max() / b.abs().max()).item(), ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
    print("pert", pert, "prod vs sep", d(outs["prod"], outs["sep"]), "| ln_fold off vs on (both sep)", d(outs["sep_nofold"], outs["sep"]))
    del unet
