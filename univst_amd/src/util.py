"""Mirror of src/util.py of the reference (same names / on-disk formats); PIL + numpy only."""
import os
import random

import numpy as np
import torch
from PIL import Image, ImageOps


def seed_everything(seed=42):
    """util.py:16-19"""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def _grid(x, nrow):
    # torchvision.utils.make_grid for the single-video case used here (b == 1 -> no padding, image unchanged)
    if x.shape[0] == 1:
        return x[0]
    import math
    b, c, h, w = x.shape
    xm = min(nrow, b)
    ym = int(math.ceil(b / xm))
    grid = x.new_zeros(c, ym * (h + 2) + 2, xm * (w + 2) + 2)
    for k in range(b):
        yy, xx = divmod(k, xm)
        grid[:, yy * (h + 2) + 2:yy * (h + 2) + 2 + h, xx * (w + 2) + 2:xx * (w + 2) + 2 + w] = x[k]
    return grid


def _frames(videos, rescale, n_rows):
    videos = videos.permute(2, 0, 1, 3, 4)            # b c t h w -> t b c h w
    for x in videos:
        x = _grid(x, n_rows).permute(1, 2, 0)
        if rescale:
            x = (x + 1.0) / 2.0
        yield (x * 255).numpy().astype(np.uint8)


def save_folder(videos: torch.Tensor, path: str, rescale=False, n_rows=4, fps=8):
    """util.py:22-31: %05d.png per frame."""
    for i, x in enumerate(_frames(videos, rescale, n_rows)):
        Image.fromarray(x.squeeze(-1) if x.shape[-1] == 1 else x).save(os.path.join(path, "%05d.png" % i))


def save_videos_grid(videos: torch.Tensor, path: str, rescale=False, n_rows=4, fps=8):
    """util.py:34-47 (mp4 through imageio when available, else an animated GIF next to the requested name)."""
    outputs = list(_frames(videos, rescale, n_rows))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    try:
        import imageio
        imageio.mimsave(path, outputs, fps=fps)
    except ImportError:
        ims = [Image.fromarray(o) for o in outputs]
        ims[0].save(os.path.splitext(path)[0] + ".gif", save_all=True, append_images=ims[1:], duration=int(1000 / fps), loop=0)


def save_images_as_mp4(images, save_path: str, fps: int = 10) -> None:
    """util.py:50-60 (10 fps there; ``fps`` lets the SD3 inversion previews keep the 8 fps of the reference's
    diffusers.utils.export_to_video call; mp4 through imageio when available, else an animated GIF next to the requested name,
    like save_videos_grid)."""
    frames = [np.array(i.convert("RGB")) for i in images]
    try:
        import imageio
        w = imageio.get_writer(save_path, fps=fps)
        for f in frames:
            w.append_data(f)
        w.close()
    except ImportError:
        ims = [Image.fromarray(f) for f in frames]
        ims[0].save(os.path.splitext(save_path)[0] + ".gif", save_all=True, append_images=ims[1:], duration=int(round(1000 / fps)), loop=0)


def load_image(image, convert_method=None, image_size=None):
    """util.py:84-120 (local paths only; there is no network on the target boxes)."""
    if not isinstance(image, str) or not os.path.isfile(image):
        raise ValueError(f"Incorrect path: {image} is not a valid path.")
    image = Image.open(image).resize(image_size)
    image = ImageOps.exif_transpose(image)
    return convert_method(image) if convert_method is not None else image.convert("RGB")


def load_video_frames(frames_path, n_frames, image_size=(512, 512)):
    """util.py:63-81"""
    frames = []
    for i in range(n_frames):
        img = load_image(f"{frames_path}/%05d.png" % i, image_size=image_size)
        if img.size != image_size:
            raise ValueError("Frame size does not match config.image_size")
        frames.append(torch.from_numpy((np.array(img) / 127.5) - 1.0).permute(2, 0, 1).float())
    return torch.stack(frames)


def load_ddim_latents_at_t(t, ddim_latents_path, is_x0=False):
    """util.py:123-130"""
    p = os.path.join(ddim_latents_path, f"ddim_x0_{t}.pt" if is_x0 else f"ddim_latents_{t}.pt")
    assert os.path.exists(p), f"Missing latents at t {t} path {p}"
    return torch.load(p, weights_only=True)


def load_mask(mask_path="", n_frames=16):
    """util.py:133-144: uint8 wrap-around multiply then clip(0,1)  =>  (pixel != 0), shape [1,F,H,W] uint8."""
    images = [np.array(Image.open(f"{mask_path}/%05d.png" % i)) * 255 for i in range(n_frames)]
    return torch.from_numpy(np.stack(images)).unsqueeze(0).clip(0, 1)
