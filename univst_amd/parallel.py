"""Frame sharding of the three-branch loop over the GPUs of one node (one process per GPU, RCCL over xGMI).

Rank r owns frames [f0, f0+Fl) of ALL THREE branches (so the PnP injection stays local).  See DESIGN.md
§multi-GPU for the exchange schedule."""
import torch


class FrameShard:
    def __init__(self, rank: int, world: int, frames: int):
        if frames % world != 0:
            raise ValueError(f"frames={frames} must be divisible by the number of GPUs ({world})")
        self.rank, self.world, self.frames = rank, world, frames
        self.local = frames // world
        self.f0 = rank * self.local

    def slice_frames(self, t: torch.Tensor) -> torch.Tensor:
        """[.., .., F, h, w] -> this rank's frames (contiguous)."""
        if self.world == 1:
            return t
        return t[:, :, self.f0:self.f0 + self.local].contiguous()

    def attach(self, unet):
        if self.world == 1:
            return
        raise NotImplementedError("multi-GPU frame sharding: see univst_amd/parallel.py")
