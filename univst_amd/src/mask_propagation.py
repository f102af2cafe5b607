"""Mirror of src/mask_propagation.py of the reference: same CLI flags, same output PNGs.

GPU work (normalise, affinity GEMM + exp, top-k threshold, column normalise, label GEMM, bilinear upsample +
per-class min-max + first-max argmax) runs in csrc/maskprop.hip; the queue bookkeeping and the
``torch.randperm`` sub-sampling stay on the host so the random index stream equals the reference's
(seed it with torch.manual_seed for reproducible masks).  The feature file is read once, not 17 times
(mask_propagation.py:104)."""
import argparse
import os
from collections import deque

import numpy as np
import torch
from PIL import Image

from .. import _native


def to_one_hot(y_tensor, n_dims=None):
    """mask_propagation.py:126-138"""
    if n_dims is None:
        n_dims = int(y_tensor.max() + 1)
    _, h, w = y_tensor.size()
    y = y_tensor.type(torch.LongTensor).view(-1, 1)
    oh = torch.zeros(y.size()[0], n_dims).scatter_(1, y, 1)
    return oh.view(h, w, n_dims).permute(2, 0, 1).unsqueeze(0).cuda()


def read_feature(path, frame_index, return_h_w=False):
    """mask_propagation.py:102-111: one frame of the dumped feature file as [h*w, C] fp32 on the GPU.  (Kept for callers of the
    reference's helper; ``propagate_masks`` reads the file ONCE instead of once per frame and queue slot.)"""
    data = torch.load(path, weights_only=True).to("cuda").float()[frame_index]
    _h, _w, _ = data.shape
    data = data.view(_h * _w, -1).contiguous()
    return (data, _h, _w) if return_h_w else data


def norm_mask(mask):
    """mask_propagation.py:114-123: per-class min-max to [0,1] for classes whose max is > 0, in place.  (Kept for callers of the
    reference's helper; the pipeline's own path fuses upsample + this + the arg-max in csrc/maskprop.hip.)"""
    mx = mask.flatten(1).max(dim=1).values
    mn = mask.flatten(1).min(dim=1).values
    sel = mx > 0
    rng = (mx - mn)[sel][:, None, None]
    mask[sel] = (mask[sel] - mn[sel][:, None, None]) / rng
    return mask


def _ws(nbytes, device):
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


def mask_propogation(feat_src_rows, feat_tar, segs, args):
    """mask_propagation.py:72-99.  Layout differs from the reference only in that source features are kept
    row-major [Nsrc, C] (the reference keeps [C, Nsrc]); returns (segs_tar, sampled feats [n, C], sampled segs)."""
    lib = _native.load()
    hw, C_ = feat_tar.shape
    Nsrc, ncls = feat_src_rows.shape[0], segs.shape[0]
    dev = feat_tar.device
    ws = _ws(lib.univst_maskprop_workspace_bytes(hw, Nsrc, C_), dev)
    segs_tar = torch.empty(ncls, hw, device=dev, dtype=torch.float32)
    _native.check(lib.univst_maskprop_frame(feat_tar.data_ptr(), feat_src_rows.data_ptr(), segs.contiguous().data_ptr(),
                                            segs_tar.data_ptr(), hw, Nsrc, C_, ncls, float(args.temperature), int(args.topk),
                                            ws.data_ptr(), _native.stream_ptr()), "maskprop_frame")
    # mask_propagation.py:87-97.  The two torch.randperm calls must see the reference's (fn, bn) in the reference's order on the
    # HOST generator (same index stream), so one number per frame has to come back from the device: fn.  Everything else stays
    # there: a stable argsort of the background flag lists the foreground positions ascending, then the background ones — what
    # torch.where(fg)[0] / torch.where(~fg)[0] return — without the two nonzero() syncs and index-list round trips of round 2.
    bg = segs_tar[0, :] == 0
    order = torch.argsort(bg.to(torch.uint8), stable=True)
    fn = hw - int(bg.sum().item())
    bn = hw - fn
    ri_f = torch.randperm(fn)[: int(fn * fn / (fn + bn) * args.sample_ratio)]
    ri_b = torch.randperm(bn)[: int(bn * bn / (fn + bn) * args.sample_ratio)]
    all_index = order[torch.cat([ri_f, ri_b + fn]).to(dev, non_blocking=True)]
    return segs_tar, feat_tar[all_index].contiguous(), segs_tar[:, all_index].contiguous()


def norm_argmax_mask(segs_tar, h, w, H, W):
    """mask_propagation.py:60-69: bilinear up, norm_mask, argmax, != 0 -> 255; uint8 [H, W] on the device."""
    lib = _native.load()
    ncls = segs_tar.shape[0]
    out = torch.empty(H, W, dtype=torch.uint8, device=segs_tar.device)
    ws = _ws(ncls * 130 * 4 + 4096, segs_tar.device)
    _native.check(lib.univst_maskprop_finalize(segs_tar.contiguous().data_ptr(), out.data_ptr(), ncls, h, w, H, W, ws.data_ptr(),
                                               _native.stream_ptr()), "maskprop_finalize")
    return out


@torch.no_grad()
def propagate_masks(features, first_mask_u8, args):
    """in-memory core of video_mask_propogation: features [F,h,w,C] (any float dtype, any device),
    first_mask_u8 numpy 'L' image -> list of F uint8 numpy masks."""
    feats = features.cuda().float()
    F_, h, w, C_ = feats.shape
    ori_h, ori_w = first_mask_u8.shape
    first = np.array(Image.fromarray(first_mask_u8).resize((w, h), 0))
    first_seg = to_one_hot(torch.from_numpy(first).float().unsqueeze(0))        # [1, ncls, h, w]
    que = deque()
    feat_first = feats[0].reshape(h * w, C_).contiguous()
    seg_first = first_seg.squeeze(0).flatten(1).contiguous()
    out = [first_mask_u8.astype(np.uint8)]
    for cnt in range(1, args.num_frames):
        feat_src = torch.cat([feat_first] + [p[0] for p in que], dim=0)
        segs_src = torch.cat([seg_first] + [p[1] for p in que], dim=-1)
        feat_tgt = feats[cnt].reshape(h * w, C_).contiguous()
        final, fs, ss = mask_propogation(feat_src, feat_tgt, segs_src, args)
        if len(que) == args.n_last_frames:
            que.popleft()
        que.append([fs, ss])
        out.append(norm_argmax_mask(final, h, w, ori_h, ori_w))
    if len(out) > 1:         # one device -> host copy for the whole clip instead of a blocking copy per frame
        rest = torch.stack(out[1:]).cpu().numpy()
        out = [out[0]] + [rest[i] for i in range(rest.shape[0])]
    return out


@torch.no_grad()
def video_mask_propogation(args):
    """mask_propagation.py:15-69"""
    name = args.mask_path.split("/")[-1].split(".")[0]
    output_path = os.path.join(args.output_path, args.backbone, name)
    os.makedirs(output_path, exist_ok=True)
    first_seg = Image.open(args.mask_path)
    feats = torch.load(args.feature_path, weights_only=True)
    masks = propagate_masks(feats, np.asarray(first_seg).astype(np.uint8), args)
    for i, m in enumerate(masks):
        Image.fromarray(m).save(os.path.join(output_path, "%05d.png" % i))
    return masks


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--temperature", default=0.2, type=float, help="The temperature for softmax.")
    parser.add_argument("--n_last_frames", type=int, default=9, help="The numbers of anchor frames.")
    parser.add_argument("--topk", type=int, default=15, help="The hyper-parameters of KNN top k.")
    parser.add_argument("--sample_ratio", type=float, default=0.3, help="The sample ratio of mask propagation.")
    parser.add_argument("--num_frames", type=int, default=16, help="The total nums of mask.")
    parser.add_argument("--mask_path", type=str, default="examples/masks/mallard-fly.png", help="The path of first frame.")
    parser.add_argument("--backbone", type=str, default=None)
    parser.add_argument("--feature_path", type=str, default=None, help="The path of inversion feature map.")
    parser.add_argument("--output_path", type=str, default=None, help="The path of output.")
    return parser


if __name__ == "__main__":
    video_mask_propogation(build_parser().parse_args())
