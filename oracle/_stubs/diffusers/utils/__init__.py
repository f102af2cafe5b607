import logging as _logging
from collections import OrderedDict
from dataclasses import fields


class BaseOutput(OrderedDict):
    """dataclass-style output with attribute + key access (the reference uses .sample and ["sample"])."""

    def __post_init__(self):
        for f in fields(self):
            v = getattr(self, f.name)
            if v is not None:
                self[f.name] = v

    def __getitem__(self, k):
        if isinstance(k, str):
            return dict(self.items())[k]
        return tuple(self.values())[k]


class _Logging:
    @staticmethod
    def get_logger(name):
        return _logging.getLogger(name)


logging = _Logging()


def deprecate(*a, **k):
    pass


def is_accelerate_available():
    return False
