"""Attribute-access dict with the behaviour the reference relies on from `easydict.EasyDict`
(pointnerf.py:134-194 builds its option tree with it; Renderer.forward returns one, renderer.py:268)."""


class AttrDict(dict):
    def __init__(self, mapping=None, **kw):
        super().__init__()
        for k, v in {**(mapping or {}), **kw}.items():
            self[k] = v

    def __setitem__(self, key, value):
        if isinstance(value, dict) and not isinstance(value, AttrDict):
            value = AttrDict(value)
        super().__setitem__(key, value)

    __setattr__ = __setitem__

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as e:
            raise AttributeError(key) from e
