"""Mirror of src/cal_optica_flow.py of the reference + the sliding-window block of
stable_diffusion.py:723-751, on the GPU (csrc/warp.hip).

RAFT (torchvision ``raft_large`` + weights) is third-party and is NOT re-implemented: every entry takes a
``flow_fn(img1_u8[H,W,3], img2_u8[H,W,3]) -> float32 [H,W,2]`` callable (device tensors).
``make_raft_flow_fn`` builds one from torchvision when it is installed (model created ONCE, not once per
call as the reference does, cal_optica_flow.py:53-55)."""
import ctypes as C

import torch

from .. import _native


def make_raft_flow_fn(device="cuda"):
    from torchvision.models.optical_flow import raft_large, Raft_Large_Weights   # third-party, optional
    model = raft_large(weights=Raft_Large_Weights.DEFAULT).to(device).eval()

    @torch.no_grad()
    def flow_fn(img1, img2):
        a = img1.permute(2, 0, 1).float().unsqueeze(0) / 255.0
        b = img2.permute(2, 0, 1).float().unsqueeze(0) / 255.0
        return model(a, b)[-1].squeeze(0).permute(1, 2, 0).contiguous()
    return flow_fn


def warp_accumulate_(acc, key, now, fwd, bwd, threshold=1.5):
    """acc[H,W,3] f32 += get_warp(key, now) (cal_optica_flow.py:51-99 with ref_image1=key, ref_image2=now)."""
    H, W, _ = key.shape
    _native.check(_native.load().univst_warp_accumulate(key.data_ptr(), now.data_ptr(), fwd.contiguous().data_ptr(),
                                                        bwd.contiguous().data_ptr(), acc.data_ptr(), H, W, float(threshold),
                                                        _native.stream_ptr()), "warp_accumulate")
    return acc


_RAFT = {}


def get_warp(image1_path, image2_path, ref_image1=None, ref_image2=None, occlusion_mask_save_path=None, warped_image_save_path=None,
             flow_fn=None):
    """cal_optica_flow.py:51-99, same positional arguments: flows between image1 and image2 (forward = image1 -> image2), occlusion
    test at 1.5 px, ``ref_image2`` warped by the forward flow and composited over ``ref_image1`` where occluded, uint8 [H,W,3].
    Images may be numpy arrays (the reference's calling convention: a numpy array comes back), uint8 device tensors (a device tensor
    comes back) or file paths.  ``flow_fn`` stands in for RAFT (third-party); when omitted the torchvision model is created ONCE per
    process (the reference re-creates it on every call, :53-55).  The ``*_save_path`` debugging outputs are written with PIL (the
    reference uses cv2.imwrite; same pixels on disk: the mask as 8-bit grey, the composite as RGB)."""
    import numpy as np

    def dev(img):
        if isinstance(img, str):
            from PIL import Image
            img = np.array(Image.open(img).convert("RGB"))
        if isinstance(img, np.ndarray):
            return torch.from_numpy(np.ascontiguousarray(img)).cuda(), True
        return img.contiguous(), False
    a, was_np = dev(image1_path)
    b, _ = dev(image2_path)
    ra = a if ref_image1 is None else dev(ref_image1)[0]
    rb = b if ref_image2 is None else dev(ref_image2)[0]
    if flow_fn is None:
        if "fn" not in _RAFT:
            _RAFT["fn"] = make_raft_flow_fn(a.device)
        flow_fn = _RAFT["fn"]
    acc = torch.zeros(*a.shape, dtype=torch.float32, device=a.device)
    fwd, bwd = flow_fn(a, b), flow_fn(b, a)
    warp_accumulate_(acc, ra, rb, fwd, bwd)
    out = acc.to(torch.uint8)
    if occlusion_mask_save_path is not None:        # cal_optica_flow.py:20-29: |fwd + bwd| > 1.5 px, pointwise
        from PIL import Image
        occ = ((fwd.float() + bwd.float()).norm(dim=-1) > 1.5).to(torch.uint8) * 255
        Image.fromarray(occ.cpu().numpy(), mode="L").save(occlusion_mask_save_path)
        print(f"Occlusion mask save at {occlusion_mask_save_path}")
    if warped_image_save_path is not None:
        from PIL import Image
        Image.fromarray(out.cpu().numpy(), mode="RGB").save(warped_image_save_path)
        print(f"Occlusion mask save at {warped_image_save_path}")      # (the reference prints this label for both files, :96)
    return out.cpu().numpy() if was_np else out


@torch.no_grad()
def sliding_window_smooth(frames, flow_fn, mask01=None, r=2):
    """stable_diffusion.py:723-751.  frames uint8 [1,3,F,H,W] (device), mask01 uint8 [F,H,W] in {0,1}
    (1 = keep the original pixel).  Gauss-Seidel over key frames like the reference (key k sees the already
    smoothed k-2, k-1), so frames are processed sequentially; each key frame is ONE launch (occlusion test + remap of
    its up to 2r neighbours + window mean, csrc/warp.hip ``warp_window_key_kernel``; 16 launches per 16-frame pass, 58 before)."""
    lib = _native.load()
    b, c, F_, H, W = frames.shape
    assert b == 1 and c == 3
    est = frames[0].permute(1, 2, 3, 0).contiguous()            # [F,H,W,3] working copy (HWC like the reference's frames)
    ori = est.clone()
    for key in range(F_):
        # the flows of this key frame against its in-clip neighbours (RAFT stand-in; estimated on the CURRENT working copy like the
        # reference: frames k-2, k-1 are already smoothed), then ONE launch: warp every neighbour, add the key frame, store the mean
        nbrs = [key + bias for bias in range(-r, r + 1) if bias != 0 and 0 <= key + bias < F_]
        key_frame = est[key]
        flows = []
        for n in nbrs:
            flows += [flow_fn(key_frame, est[n]).to(torch.float32).contiguous(), flow_fn(est[n], key_frame).to(torch.float32).contiguous()]
        ptrs = (C.c_void_p * max(1, len(flows)))(*[f.data_ptr() for f in flows])
        _native.check(lib.univst_warp_window_key(est.data_ptr(), ptrs, len(nbrs), F_, H, W, key, r, 1.5, _native.stream_ptr()), "warp_window_key")
    if mask01 is not None:
        m = mask01.to(torch.bool)[..., None]
        est = torch.where(m, ori, est)
    return est.permute(3, 0, 1, 2).unsqueeze(0).contiguous()


# ------------------------------------------------------------------------------------------------ latent-space variant
@torch.no_grad()
def make_latent_flows(frames_u8, flow_fn, r=2, down=8):
    """flows of the CONTENT clip at latent resolution, computed once before the loop (the motion the smoother follows is the
    content's; the pixel variant re-estimates it from every decoded x0: 116 RAFT inferences per step, stable_diffusion.py:743).
    frames_u8 [F,H,W,3] uint8 (device) -> lflow [F, 2r+1, H/down, W/down, 2] fp32: flow from frame k to frame k+b averaged
    over down x down pixel blocks and divided by `down` (latent-pixel units); slots with k+b outside the clip stay zero."""
    F_, H, W, _ = frames_u8.shape
    out = torch.zeros(F_, 2 * r + 1, H // down, W // down, 2, dtype=torch.float32, device=frames_u8.device)
    for k in range(F_):
        for b in range(-r, r + 1):
            n = k + b
            if b == 0 or n < 0 or n >= F_:
                continue
            f = flow_fn(frames_u8[k], frames_u8[n]).to(torch.float32)                        # [H,W,2]
            f = torch.nn.functional.avg_pool2d(f.permute(2, 0, 1)[None], down)[0].permute(1, 2, 0) / down
            out[k, b + r] = f
    return out


@torch.no_grad()
def latent_sliding_window_smooth(x0, lflow, mask_m=None, r=2, threshold=1.5 / 8):
    """latent-space sliding window (README.md:59 of the reference; our definition, see csrc/warp.hip):
    x0 [1,C,F,h,w] fp16 (pred_original_sample), lflow from make_latent_flows, mask_m fp16 [F,h,w] (1 keeps the un-smoothed
    latent, the polarity of stable_diffusion.py:751).  Returns the smoothed x0 (new tensor)."""
    b, C_, F_, h, w = x0.shape
    assert b == 1 and tuple(lflow.shape) == (F_, 2 * r + 1, h, w, 2)
    est = x0.to(torch.float16).contiguous().clone()
    _native.check(_native.load().univst_latent_window_smooth(est.data_ptr(), lflow.contiguous().data_ptr(), C_, F_, h, w, r,
                                                             float(threshold), _native.stream_ptr()), "latent_window_smooth")
    if mask_m is not None:
        est = _native.mask_blend(est, x0.to(torch.float16).contiguous(), mask_m)
    return est
