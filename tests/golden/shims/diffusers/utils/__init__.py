import logging as _pylogging
from collections import OrderedDict
from dataclasses import fields


def deprecate(*args, **kwargs):
    return None


class _Logging:
    @staticmethod
    def get_logger(name):
        return _pylogging.getLogger(name)


logging = _Logging()


class BaseOutput(OrderedDict):
    """dataclass-style output with attribute and key access (what the reference's callers use: `.sample`, `[0]`)"""

    def __post_init__(self):
        for f in fields(self):
            v = getattr(self, f.name)
            if v is not None:
                self[f.name] = v

    def __getitem__(self, k):
        if isinstance(k, str):
            return dict(self.items())[k]
        return self.to_tuple()[k]

    def to_tuple(self):
        return tuple(self[k] for k in self.keys())
