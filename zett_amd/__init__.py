"""zett_amd — MI355X-native implementation of ZeTT's embedding-prediction hot path.

Importing the package registers the hypernetwork with the transformers Auto
classes, so ``AutoModel.from_pretrained(<zett checkpoint>)`` returns the HIP-backed
``ZettHypernet`` (the reference registers its torch port the same way in
scripts/convert_to_pt.py:26-27).
"""
import os as _os


def configure_hw_queues(world_size=None, log=True) -> bool:
    """Launcher hook (bench.py, scripts/transfer.py call it first thing; importing the package does NOT): one process per GPU
    under torchrun (WORLD_SIZE > 1) wants GPU_MAX_HW_QUEUES=8.  The row exchange is real xGMI traffic there, and it — like the
    plan-ahead of zett_forward_prepare — overlaps a forward only if its streams do not share the forward's HARDWARE queue; the
    HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES (default 4) queues and an eagerly initialised RCCL
    communicator takes its share first (profiles/r4g_blocks.md, NOTEBOOK R4.9).  The variable is read when the HIP runtime
    initialises (the first CUDA call of the process): call this BEFORE any torch.cuda call; a value the user set wins; a single
    process keeps the default (its "exchange" is a local copy and serialised streams are the faster schedule).  Returns True
    when the variable was set here."""
    world = int(world_size if world_size is not None else (_os.environ.get("WORLD_SIZE", "1") or 1))
    if world <= 1 or "GPU_MAX_HW_QUEUES" in _os.environ:
        return False
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"
    if log and int(_os.environ.get("RANK", "0") or 0) == 0:
        import sys
        print(f"[zett_amd] WORLD_SIZE={world}: GPU_MAX_HW_QUEUES=8 for this process (set it yourself to override)", file=sys.stderr)
    return True


from .config import MODEL_TYPE, ZettHypernetConfig  # noqa: F401,E402


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
