import functools
import inspect


class FrozenDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


class ConfigMixin:
    @property
    def config(self):
        return self._internal_dict

    def register_to_config(self, **kw):
        d = dict(getattr(self, "_internal_dict", {}))
        d.update(kw)
        self._internal_dict = FrozenDict(d)


def register_to_config(init):
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        sig = inspect.signature(init)
        cfg = {}
        params = list(sig.parameters.items())[1:]
        for (name, p) in params:
            if p.kind in (p.VAR_KEYWORD, p.VAR_POSITIONAL):
                continue
            if p.default is not inspect.Parameter.empty:
                cfg[name] = p.default
        for (name, _), a in zip(params, args):
            cfg[name] = a
        cfg.update(kwargs)
        init(self, *args, **kwargs)
        self._internal_dict = FrozenDict(cfg)
    return inner
