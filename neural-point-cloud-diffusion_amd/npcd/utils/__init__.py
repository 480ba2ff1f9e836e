from .attrdict import AttrDict  # noqa: F401
