import contextlib
import torch


class _Bar:
    def update(self, n=1):
        pass


class DiffusionPipeline:
    def __init__(self):
        pass

    def register_modules(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def device(self):
        for m in vars(self).values():
            if isinstance(m, torch.nn.Module):
                for p in m.parameters():
                    return p.device
        return torch.device("cpu")

    @contextlib.contextmanager
    def progress_bar(self, iterable=None, total=None):
        yield _Bar()

    @staticmethod
    def numpy_to_pil(images):
        raise NotImplementedError
