"""Flax msgpack reader / name map: round trip through the in-repo writer, and a checkpoint
assembled by hand in flax's on-disk layout with the Flax-side parameter names (tests/flax_fixture.py)."""
import os

import numpy as np

from zett_amd import synth
from zett_amd.dims import weight_shapes
from zett_amd.flax_io import load_flax_checkpoint, read_msgpack, torch_state_dict_to_flax, write_msgpack


def test_round_trip_and_name_map(tmp_path):
    cfg, *_ = synth.workload("tiny")
    w = synth.make_weights(cfg, 4)
    tree = torch_state_dict_to_flax(w)
    # spot-check the Flax-side layout the reference uses (zett/model/__init__.py, convert_to_pt.py:35-46)
    assert tree["input_projection"]["layers_0"]["kernel"].shape == w["input_projection.0.weight"].T.shape
    assert "scale" in tree["input_projection"]["layers_1"]["ln"]
    assert "embedding" in tree["model"]["embeddings"]["lang_embedding"]
    assert tree["model"]["encoder"]["layer"]["1"]["attention"]["self"]["query"]["kernel"].shape == (128, 128)
    path = os.path.join(tmp_path, "flax_model.msgpack")
    write_msgpack(tree, path)
    back = load_flax_checkpoint(path)
    expected = {k: v for k, v in w.items() if k != "model.embeddings.word_embeddings.weight"}
    assert set(back) == set(expected)
    for k, v in expected.items():
        np.testing.assert_array_equal(back[k], v, err_msg=k)
    assert set(weight_shapes(cfg)) - set(back) == {"model.embeddings.word_embeddings.weight"}


def test_bfloat16_and_chunked_arrays(tmp_path):
    import msgpack
    a = np.arange(12, dtype=np.float32).reshape(3, 4)
    bf = (a.view(np.uint32) >> 16).astype(np.uint16)
    blob = msgpack.packb({"x": msgpack.ExtType(1, msgpack.packb(([3, 4], "bfloat16", bf.tobytes()), use_bin_type=True)),
                          "y": {"__msgpack_chunked_array__": True, "shape": [2, 3],
                                "chunks": {"0": msgpack.ExtType(1, msgpack.packb(([4], "float32", a.reshape(-1)[:4].tobytes()), use_bin_type=True)),
                                           "1": msgpack.ExtType(1, msgpack.packb(([2], "float32", a.reshape(-1)[4:6].tobytes()), use_bin_type=True))}}},
                         use_bin_type=True)
    p = os.path.join(tmp_path, "t.msgpack")
    open(p, "wb").write(blob)
    t = read_msgpack(p)
    np.testing.assert_array_equal(t["x"], a)             # small integers are exact in bfloat16
    np.testing.assert_array_equal(t["y"], a.reshape(-1)[:6].reshape(2, 3))


def test_hypernet_from_flax_checkpoint(tmp_path):
    import torch

    from zett_amd.config import ZettHypernetConfig
    from zett_amd.hypernet import ZettHypernet
    cfg, *_ = synth.workload("tiny")
    w = synth.make_weights(cfg, 6)
    ZettHypernetConfig(**cfg).save_pretrained(tmp_path)
    write_msgpack(torch_state_dict_to_flax(w), os.path.join(tmp_path, "flax_model.msgpack"))
    model = ZettHypernet.from_flax_checkpoint(str(tmp_path))
    for k, v in w.items():
        if k != "model.embeddings.word_embeddings.weight":
            assert torch.equal(model.state_dict()[k], torch.from_numpy(v)), k


def test_hand_assembled_flax_checkpoint(tmp_path):
    """Bytes laid out as flax.serialization writes them (ExtType 1 leaves, a bfloat16 leaf, a chunked array whose shape
    is a {"0": .., "1": ..} dict), names as the Flax modules spell them: the loader must return the PyTorch state dict."""
    import torch

    from tests.flax_fixture import write_flax_checkpoint
    from zett_amd.config import ZettHypernetConfig
    from zett_amd.hypernet import ZettHypernet
    cfg, *_ = synth.workload("tiny")
    w = synth.make_weights(cfg, 8)
    ZettHypernetConfig(**cfg).save_pretrained(tmp_path)
    expected = write_flax_checkpoint(str(tmp_path), cfg, w)
    model = ZettHypernet.from_flax_checkpoint(str(tmp_path))
    state = model.state_dict()
    assert set(expected) | {"model.embeddings.word_embeddings.weight"} >= set(state) - {"model.embeddings.word_embeddings.weight"}
    for k, v in expected.items():
        assert torch.equal(state[k], torch.from_numpy(v)), k
    # the bfloat16 leaf really was stored in two bytes per element, the chunked one in pieces
    import msgpack
    raw = msgpack.unpackb(open(os.path.join(tmp_path, "flax_model.msgpack"), "rb").read(), raw=False, strict_map_key=False)
    shape, dtype_name, buf = msgpack.unpackb(raw["model"]["embeddings"]["position_embeddings"]["embedding"].data, raw=False)
    assert dtype_name == "bfloat16" and len(buf) == 2 * int(np.prod(shape))
    assert raw["fallback_embeddings"]["embedding"]["__msgpack_chunked_array__"] and isinstance(raw["fallback_embeddings"]["embedding"]["shape"], dict)
