"""The BSON container of scope row f-4 (latentdiffeq_amd/bson.py; host code — CPU only).

Pinned by what is published: the BSON 1.1 specification's own example documents (bsonspec.org, "Specification" page:
{"hello": "world"} and {"BSON": ["awesome", 5.05, 1986]}) byte for byte, hand-assembled documents for every element type the
container uses, and BSON.jl's published lowering of tuples and arrays (tag / type / size / data). No file written by BSON.jl exists in
the reference tree, so the Julia side is unpinned (stated in the module header)."""
import struct

import numpy as np
import pytest

from latentdiffeq_amd import bson, data as D


def test_specification_examples_byte_for_byte():
    hello = b"\x16\x00\x00\x00\x02hello\x00\x06\x00\x00\x00world\x00\x00"
    assert bson.dumps({"hello": "world"}) == hello and bson.loads(hello) == {"hello": "world"}
    # the second example stores 1986 as int32 (0x10); this writer emits Julia's Int as int64, so it is checked on the reading side
    awesome = (b"\x31\x00\x00\x00\x04BSON\x00\x26\x00\x00\x00\x020\x00\x08\x00\x00\x00awesome\x00\x011\x00\x33\x33\x33\x33\x33\x33\x14\x40"
               b"\x102\x00\xc2\x07\x00\x00\x00\x00")
    assert bson.loads(awesome) == {"BSON": ["awesome", 5.05, 1986]}


def test_every_element_type_round_trips_and_has_the_specified_layout():
    doc = {"d": 1.5, "s": "π", "o": {"k": None}, "a": [True, False], "b": b"\x01\x02\x03", "i": 7, "n": None}
    raw = bson.dumps(doc)
    assert struct.unpack_from("<i", raw)[0] == len(raw) and raw[-1] == 0
    assert bson.loads(raw) == doc
    assert b"\x05b\x00\x03\x00\x00\x00\x00\x01\x02\x03" in raw             # binary: int32 length, subtype 0x00, bytes
    assert b"\x12i\x00\x07\x00\x00\x00\x00\x00\x00\x00" in raw             # Int → int64
    assert b"\x04a\x00" in raw and b"\x080\x00\x01\x081\x00\x00" in raw    # array = document with keys "0", "1", …
    for bad in (raw[:-1], raw + b"\x00", b"\x05\x00\x00\x00\x01", raw[:4] + b"\x7f" + raw[5:]):
        with pytest.raises(ValueError):
            bson.loads(bad)


def test_julia_lowering_of_arrays_and_tuples():
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    low = bson.lower(a)
    assert low["tag"] == "array" and low["type"] == {"tag": "datatype", "name": ["Core", "Float32"], "params": []} and low["size"] == [2, 3]
    assert np.frombuffer(low["data"], np.float32).tolist() == [0, 3, 1, 4, 2, 5]          # column-major, as Julia stores it
    t = bson.lower((a, [a, a]))
    assert t["tag"] == "tuple" and t["data"][1]["type"]["name"] == ["Core", "Array"] and t["data"][1]["type"]["params"][1] == 2
    back = bson.raise_(bson.loads(bson.dumps({"x": t})))["x"]
    assert isinstance(back, tuple) and np.array_equal(back[0], a) and np.array_equal(back[1][1], a)


def test_dataset_file_in_the_reference_nesting(tmp_path):
    rng = np.random.default_rng(0)
    n, T = 5, 7
    latent = rng.standard_normal((2, T, n)).astype(np.float32)
    u0s, ps = rng.standard_normal((2, n)).astype(np.float32), rng.uniform(1, 2, (1, n)).astype(np.float32)
    high = rng.uniform(0, 1, (28, 28, T, n)).astype(np.float32)
    path = D.save_dataset(str(tmp_path / "data.bson"), latent, u0s, ps, high)
    doc = bson.load(path)
    lat_j, u0_j, ps_j, high_j = doc["data"]                                   # data = (latent_data, u0s, ps, high_dim_data)  [REF create_data.jl:57]
    assert len(lat_j) == n and lat_j[0].shape == (2, T) and u0_j[0].shape == (2,) and ps_j[0].shape == (1, 1)
    assert len(high_j) == n and len(high_j[0]) == T and high_j[0][0].shape == (28, 28)
    assert np.array_equal(high_j[3][2], high[:, :, 2, 3])
    got = D.load_dataset(path)
    for g, w in zip(got, (latent, u0s, ps, high)):
        assert g.dtype == np.float32 and np.array_equal(g, w)
    prep = D.prepare_training_data(got)                                       # the example script's next step works on it unchanged
    assert prep["train_set"].shape[0] == 784 and prep["full_seq_len"] == T
    with pytest.raises(KeyError):
        bson.save(str(tmp_path / "other.bson"), weights=[u0s])
        D.load_dataset(str(tmp_path / "other.bson"))


def test_weights_file(tmp_path):
    ws = [np.arange(5, dtype=np.float32), np.ones(3, np.float32)]
    got = D.load_weights(D.save_weights(str(tmp_path / "best_model_weights.bson"), ws))
    assert len(got) == 2 and all(np.array_equal(a, b) for a, b in zip(got, ws))
