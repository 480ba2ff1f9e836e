"""Model classes of the reference's `npcd.models` package (npcd/models/__init__.py exports NPCD)."""


def __getattr__(name):          # lazy: importing npcd.models.diffusion must not pull in the renderer
    if name == "NPCD":
        from .npcd import NPCD
        return NPCD
    raise AttributeError(name)
