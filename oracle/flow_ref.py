"""Oracle: optical-flow warp, occlusion test and sliding-window smoothing (numpy restatement).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
Restates src/cal_optica_flow.py:20-99 and the window loop of
backbones/video_diffusion_sd/pipelines/stable_diffusion.py:725-751.  RAFT (torchvision) is third-party and
stays outside: a ``flow_fn(img1_u8[H,W,3], img2_u8[H,W,3]) -> float32 [H,W,2]`` callable stands in.
``cv2.remap`` (OpenCV 4.9.0, absent here => parity unpinned) is restated from its published fixed-point
bilinear algorithm: INTER_BITS=5 sub-pixel positions, INTER_REMAP_COEF_BITS=15 weights,
``(sum + 2^14) >> 15`` rounding, BORDER_CONSTANT(0) taps.
"""
from typing import Callable, Optional

import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
COEF_BITS = 15


def compute_occlusion_mask(fwd: np.ndarray, bwd: np.ndarray, threshold=1.0) -> np.ndarray:
    """cal_optica_flow.py:20-29 (float32 arithmetic in the reference's operation order)."""
    h, w, _ = fwd.shape
    gx, gy = np.meshgrid(np.arange(w), np.arange(h))
    c2 = np.stack([gx, gy], axis=-1).astype(np.float32)
    c1 = c2 + fwd
    back = c1 + bwd
    err = np.linalg.norm(back - c2, axis=-1)
    return (err > threshold).astype(np.uint8) * 255


def remap_bilinear_u8(image: np.ndarray, map_x: np.ndarray, map_y: np.ndarray) -> np.ndarray:
    """cv2.remap(image, map_x, map_y, INTER_LINEAR, BORDER_CONSTANT) for uint8 HxWxC, float32 maps."""
    H, W = image.shape[:2]
    sx = np.rint(map_x.astype(np.float32) * np.float32(INTER_TAB_SIZE)).astype(np.int64)
    sy = np.rint(map_y.astype(np.float32) * np.float32(INTER_TAB_SIZE)).astype(np.int64)
    ix = np.clip(sx >> INTER_BITS, -32768, 32767)       # saturate_cast<short>
    iy = np.clip(sy >> INTER_BITS, -32768, 32767)
    fx = (sx & (INTER_TAB_SIZE - 1)).astype(np.int64)
    fy = (sy & (INTER_TAB_SIZE - 1)).astype(np.int64)
    s = 1 << (COEF_BITS - 2 * INTER_BITS)                # 32: (32-fx)(32-fy)*32 sums to 2^15 exactly
    w00 = (INTER_TAB_SIZE - fx) * (INTER_TAB_SIZE - fy) * s
    w01 = fx * (INTER_TAB_SIZE - fy) * s
    w10 = (INTER_TAB_SIZE - fx) * fy * s
    w11 = fx * fy * s
    img = image.astype(np.int64)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
        return v * ok[..., None]

    acc = (tap(iy, ix) * w00[..., None] + tap(iy, ix + 1) * w01[..., None]
           + tap(iy + 1, ix) * w10[..., None] + tap(iy + 1, ix + 1) * w11[..., None])
    return np.clip((acc + (1 << (COEF_BITS - 1))) >> COEF_BITS, 0, 255).astype(np.uint8)


def warp_image_with_flow(image: np.ndarray, flow: np.ndarray) -> np.ndarray:
    """cal_optica_flow.py:31-41."""
    h, w, _ = flow.shape
    gx, gy = np.meshgrid(np.arange(w), np.arange(h))
    c2 = np.stack([gx, gy], axis=-1).astype(np.float32)
    c1 = c2 + flow
    return remap_bilinear_u8(image, c1[..., 0].astype(np.float32), c1[..., 1].astype(np.float32))


def apply_mask(image: np.ndarray, mask: np.ndarray, original: np.ndarray) -> np.ndarray:
    """cal_optica_flow.py:43-46."""
    m = np.repeat(mask[:, :, np.newaxis], 3, axis=2) / 255.0
    return (image * (1 - m) + original * m).astype(np.uint8)


def get_warp(flow_fn: Callable, image1: np.ndarray, image2: np.ndarray) -> np.ndarray:
    """cal_optica_flow.py:51-99 with ref_image1=image1, ref_image2=image2 (how the pipeline calls it,
    stable_diffusion.py:743) and the RAFT model replaced by ``flow_fn``."""
    fwd = flow_fn(image1, image2)
    bwd = flow_fn(image2, image1)
    occ = compute_occlusion_mask(fwd, bwd, threshold=1.5)
    warped = warp_image_with_flow(image2, fwd)
    return apply_mask(warped, occ, image1)


def sliding_window_smooth(frames: np.ndarray, flow_fn: Callable, mask01: Optional[np.ndarray], r: int = 2):
    """stable_diffusion.py:723-751.  frames uint8 [1,3,F,H,W] (modified Gauss-Seidel style, in place on a
    copy), mask01 uint8 [1,F,H,W] in {0,1} (1 = keep original).  Returns uint8 [1,3,F,H,W]."""
    est = frames.copy()
    ori = frames.copy()
    nf = est.shape[2]
    tmp = np.zeros_like(est).astype(np.float32)
    for key in range(nf):
        key_frame = est[:, :, key][0].transpose(1, 2, 0).copy()
        weight = 0
        for bias in range(-r, r + 1):
            now = key + bias
            if 0 <= now < nf:
                now_frame = est[:, :, now][0].transpose(1, 2, 0).copy()
                if bias == 0:
                    tmp[:, :, key] = tmp[:, :, key] + now_frame.transpose(2, 0, 1).astype(np.float32)
                else:
                    wr = get_warp(flow_fn, key_frame, now_frame).transpose(2, 0, 1)
                    tmp[:, :, key] = tmp[:, :, key] + wr.astype(np.float32)
                weight += 1
        est[:, :, key] = tmp[:, :, key] / weight          # float32 -> uint8 store truncates
    est = est.astype(np.uint8)
    if mask01 is not None:
        m = mask01[None, :].astype(np.uint8)
        est = ori * m + (1 - m) * est
    return est


def latent_sliding_window_smooth(x0: np.ndarray, lflow: np.ndarray, mask_m: Optional[np.ndarray], r: int = 2,
                                 threshold: float = 1.5 / 8) -> np.ndarray:
    """Latent-space sliding window.  NOT a restatement of reference code — the reference only describes it
    (README.md:59) and implements the pixel variant; this is the definition univst_amd ships behind smoother='latent',
    written out independently of the kernel for the parity test: x0 [1,C,F,h,w] float32, lflow [F,2r+1,h,w,2] (flow from
    frame k to k+b in latent pixels), mask_m [F,h,w] in [0,1] (1 keeps the un-smoothed latent)."""
    est = x0.astype(np.float32).copy()
    ori = est.copy()
    _, C, F, h, w = est.shape
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    for key in range(F):
        acc = est[0, :, key].copy()
        weight = 1.0
        for b in range(-r, r + 1):
            now = key + b
            if b == 0 or now < 0 or now >= F:
                continue
            fwd, bwd = lflow[key, b + r], lflow[now, r - b]
            e = fwd + bwd
            occ = np.sqrt(e[..., 0] ** 2 + e[..., 1] ** 2) > threshold
            sx, sy = xs + fwd[..., 0], ys + fwd[..., 1]
            ix, iy = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
            ax, ay = sx - ix, sy - iy
            src = est[0, :, now]

            def tap(yy, xx):
                ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
                return src[:, np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)] * ok[None]
            warped = (tap(iy, ix) * ((1 - ax) * (1 - ay))[None] + tap(iy, ix + 1) * (ax * (1 - ay))[None]
                      + tap(iy + 1, ix) * ((1 - ax) * ay)[None] + tap(iy + 1, ix + 1) * (ax * ay)[None])
            acc = acc + np.where(occ[None], est[0, :, key], warped)
            weight += 1.0
        est[0, :, key] = (acc / weight).astype(np.float16).astype(np.float32)      # the product stores fp16
    if mask_m is not None:
        m = mask_m[None, None].astype(np.float32)
        est = (1 - m) * est + m * ori
    return est
