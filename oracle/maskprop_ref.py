"""Oracle: point-matching mask propagation (in-memory restatement).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
Restates src/mask_propagation.py:15-138 of the reference.  File I/O is lifted out: features come in as a
[F,h,w,C] tensor (the ``inversion_feature_map_2_block_301_step.pt`` payload) and masks come back as uint8
arrays.  Sampling uses ``torch.randperm`` on the global CPU RNG exactly like the reference
(mask_propagation.py:92-95): seed it with ``torch.manual_seed`` for reproducible / bit-exact runs.
"""
from collections import deque
from typing import List

import numpy as np
import torch
import torch.nn.functional as F


def to_one_hot(seg: torch.Tensor) -> torch.Tensor:
    """mask_propagation.py:126-138: [1,h,w] float labels -> [1, max+1, h, w] one-hot (up to 256 classes)."""
    n_dims = int(seg.max() + 1)
    _, h, w = seg.shape
    y = seg.long().view(-1, 1)
    oh = torch.zeros(y.shape[0], n_dims).scatter_(1, y, 1)
    return oh.view(h, w, n_dims).permute(2, 0, 1).unsqueeze(0)


def norm_mask(mask: torch.Tensor) -> torch.Tensor:
    """mask_propagation.py:114-123: per-class min-max to [0,1] for classes whose max > 0 (in place)."""
    for c in range(mask.shape[0]):
        m = mask[c]
        if m.max() > 0:
            m = m - m.min()
            m = m / m.max()
            mask[c] = m
    return mask


def mask_propogation(feat_src, feat_tar, segs, temperature=0.2, topk=15, sample_ratio=0.3):
    """mask_propagation.py:72-99.  feat_src [C,Nsrc], feat_tar [hw,C], segs [ncls,Nsrc]."""
    feat_tar_ori = feat_tar.T
    feat_src = F.normalize(feat_src, dim=0, p=2)
    feat_tar = F.normalize(feat_tar, dim=1, p=2)
    aff = torch.exp(feat_tar @ feat_src / temperature).transpose(1, 0)       # [Nsrc, hw]
    tk_val_min = torch.topk(aff, topk, dim=0).values.min(dim=0).values
    aff[aff < tk_val_min] = 0
    aff = aff / torch.sum(aff, keepdim=True, axis=0)
    segs_tar = torch.mm(segs, aff)
    fore_index = torch.where(segs_tar[0, :] != 0)[0]
    back_index = torch.where(segs_tar[0, :] == 0)[0]
    fn, bn = len(fore_index), len(back_index)
    ri = torch.randperm(fn)[: int(fn * fn / (fn + bn) * sample_ratio)]
    fs = fore_index[ri]
    ri = torch.randperm(bn)[: int(bn * bn / (fn + bn) * sample_ratio)]
    bs = back_index[ri]
    all_index = torch.cat([fs, bs])
    return segs_tar, feat_tar_ori[:, all_index], segs_tar[:, all_index]


def video_mask_propagation(features: torch.Tensor, first_mask_u8: np.ndarray, num_frames=16, n_last_frames=9,
                           temperature=0.2, topk=15, sample_ratio=0.3, return_soft=False) -> List[np.ndarray]:
    """mask_propagation.py:15-69.  ``features`` [F,h,w,C] (any float dtype; upcast like read_feature :104),
    ``first_mask_u8`` the 'L' first-frame mask at full resolution.  Returns masks for frames 0..F-1
    (frame 0 = the input verbatim, :29)."""
    from PIL import Image
    features = features.float()
    _, h, w, _ = features.shape
    ori_h, ori_w = first_mask_u8.shape
    first = np.array(Image.fromarray(first_mask_u8).resize((w, h), 0))
    first_seg = to_one_hot(torch.from_numpy(first).float().unsqueeze(0))
    que = deque()
    feat_first = features[0].reshape(h * w, -1).T
    out = [first_mask_u8.astype(np.uint8)]
    soft = []
    for cnt in range(1, num_frames):
        feat_src = torch.cat([feat_first] + [p[0] for p in que], dim=-1)
        segs_src = torch.cat([first_seg.squeeze(0).flatten(1)] + [p[1] for p in que], dim=-1)
        C = segs_src.shape[0]
        feat_tgt = features[cnt].reshape(h * w, -1)
        final, feat_s, segs_s = mask_propogation(feat_src, feat_tgt, segs_src, temperature, topk, sample_ratio)
        if len(que) == n_last_frames:
            que.popleft()
        que.append([feat_s, segs_s])
        soft.append(final.clone())
        final = final.reshape(1, C, h, w)
        final = F.interpolate(final, size=(ori_h, ori_w), mode="bilinear", align_corners=False)[0]
        final = norm_mask(final)
        _, idx = torch.max(final, dim=0)
        m = idx.numpy().astype(np.uint8)
        m[m != 0] = 255
        out.append(m)
    return (out, soft) if return_soft else out
