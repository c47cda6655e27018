import functools
import inspect


class FrozenDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


class ConfigMixin:
    config_name = None

    def register_to_config(self, **kwargs):
        # diffusers <= 0.15 also mirrors every config entry as a plain attribute (`unet.in_channels`, which the reference's
        # pipeline reads at pipeline_stable_diffusion_e4t.py:163)
        for k, v in kwargs.items():
            try:
                object.__setattr__(self, k, v)
            except AttributeError:
                pass
        object.__setattr__(self, "_internal_dict", FrozenDict({**self.__dict__.get("_internal_dict", {}), **kwargs}))

    @property
    def config(self):
        return self._internal_dict


def register_to_config(init):
    """records every __init__ argument (defaults included) in `self.config`, like diffusers' decorator"""
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        sig = inspect.signature(init)
        params = [p for n, p in sig.parameters.items() if n != "self" and p.kind not in (p.VAR_KEYWORD, p.VAR_POSITIONAL)]
        cfg = {p.name: p.default for p in params}
        for p, a in zip(params, args):
            cfg[p.name] = a
        cfg.update({k: v for k, v in kwargs.items() if not k.startswith("_")})
        ConfigMixin.register_to_config(self, **cfg)
        init(self, *args, **{k: v for k, v in kwargs.items() if not k.startswith("_")})
    return inner
