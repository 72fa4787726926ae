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


def install_remote_code(checkpoint_dir: str) -> None:
    """Make ``AutoModel.from_pretrained(checkpoint_dir, trust_remote_code=True)`` return the HIP-backed
    hypernetwork for a checkpoint written by the reference.

    Reference checkpoints carry ``auto_map`` entries that point at ``configuration_hypernet.py`` /
    ``modeling_hypernet.py`` inside the checkpoint directory (scripts/convert_to_pt.py:26-27,49), i.e. at
    the reference's own torch implementation.  This writes two shim modules of those names that re-export
    the zett_amd classes, leaving weights and config untouched.
    """
    import json
    import os

    shims = {
        "configuration_hypernet.py": "from zett_amd.config import ZettHypernetConfig  # noqa: F401\n",
        "modeling_hypernet.py": "from zett_amd.hypernet import ZettHypernet  # noqa: F401\n",
    }
    for name, body in shims.items():
        with open(os.path.join(checkpoint_dir, name), "w") as f:
            f.write('"""Shim written by zett_amd.install_remote_code: routes the checkpoint\'s auto_map to zett_amd."""\n' + body)
    cfg_path = os.path.join(checkpoint_dir, "config.json")
    with open(cfg_path) as f:
        cfg = json.load(f)
    cfg["auto_map"] = {"AutoConfig": "configuration_hypernet.ZettHypernetConfig", "AutoModel": "modeling_hypernet.ZettHypernet"}
    with open(cfg_path, "w") as f:
        json.dump(cfg, f, indent=2)
