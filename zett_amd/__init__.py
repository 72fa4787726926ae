"""zett_amd — MI355X-native implementation of ZeTT's embedding-prediction hot path.

Importing the package registers the hypernetwork with the transformers Auto
classes, so ``AutoModel.from_pretrained(<zett checkpoint>)`` returns the HIP-backed
``ZettHypernet`` (the reference registers its torch port the same way in
scripts/convert_to_pt.py:26-27).
"""
from .config import MODEL_TYPE, ZettHypernetConfig  # noqa: F401


def _register() -> None:
    from transformers import AutoConfig, AutoModel

    from .hypernet import ZettHypernet

    try:
        AutoConfig.register(MODEL_TYPE, ZettHypernetConfig)
    except ValueError:
        pass
    try:
        AutoModel.register(ZettHypernetConfig, ZettHypernet)
    except ValueError:
        pass


def __getattr__(name):
    if name == "ZettHypernet":
        from .hypernet import ZettHypernet
        return ZettHypernet
    if name == "get_surface_form_matrix":
        from .surface_forms import get_surface_form_matrix
        return get_surface_form_matrix
    raise AttributeError(name)


_register()
