"""BSON container for the dataset and weight files of scope row f-4 (host code, no kernels).

The reference keeps its data set and its best weights in BSON files written by BSON.jl (`@save data_path data`,
`@save ".../best_model_weights.bson" weights`) [REF examples/pendulum_friction-less/model_train.jl:86-95, :215]. BSON.jl is an
un-vendored dependency (Project.toml: BSON = "0.2, 0.3") and no file written by it exists in the reference tree, so this module is
**unpinned** against reference-produced bytes. What it follows:

  * the container: BSON 1.1 (bsonspec.org) — a document is int32 length · elements · 0x00; an element is a type byte, a C-string
    key and a value; types used: 0x01 double, 0x02 string, 0x03 document, 0x04 array, 0x05 binary (subtype 0x00), 0x08 bool,
    0x0A null, 0x10 int32, 0x12 int64. Pinned by the specification's own examples (tests/test_bson.py).
  * the lowering of Julia values to such documents as BSON.jl publishes it (its README, "Notes" / src/extensions.jl): a `Tuple` is
    {tag: "tuple", data: [...]}; an `Array{T,N}` of a bits type is {tag: "array", type: {tag: "datatype", name: ["Core", "Float32"],
    params: []}, size: [d1, …, dN], data: <the bytes, column-major>}; an array of arrays carries its elements, lowered, in `data`
    and its element type as {tag: "datatype", name: ["Core", "Array"], params: [<eltype>, N]}; `@save f x` writes the document
    {x: lower(x)}.

`dump_dataset` / `load_dataset` write and read the reference's `data = (latent_data, u0s, ps, high_dim_data)` in its own nesting —
latent_data :: Vector{Matrix{Float32}} (2 × T each), u0s :: Vector{Vector{Float32}}, ps :: Vector{Matrix{Float32}} (1 × 1),
high_dim_data :: Vector{Vector{Matrix{Float32}}} (28 × 28 frames) [REF examples/pendulum_friction-less/create_data.jl:31-57,
:100-117] — from / to the dense arrays the rest of this package uses (latent [2, T, n], u0s [2, n], ps [1, n], frames [28, 28, T, n])."""
import struct

import numpy as np

_JL = {np.dtype(np.float32): "Float32", np.dtype(np.float64): "Float64", np.dtype(np.int64): "Int64", np.dtype(np.int32): "Int32",
       np.dtype(np.uint8): "UInt8"}
_NP = {v: k for k, v in _JL.items()}


# ---- the container ---------------------------------------------------------------------------------------------------
def _cstr(s: str) -> bytes:
    b = s.encode("utf-8")
    if b"\x00" in b:
        raise ValueError("BSON keys cannot contain NUL")
    return b + b"\x00"


def _elem(key: str, v) -> bytes:
    k = _cstr(key)
    if isinstance(v, bool):
        return b"\x08" + k + (b"\x01" if v else b"\x00")
    if v is None:
        return b"\x0a" + k
    if isinstance(v, (int, np.integer)):
        return b"\x12" + k + struct.pack("<q", int(v))                    # Julia's Int is Int64: BSON.jl writes 0x12
    if isinstance(v, (float, np.floating)):
        return b"\x01" + k + struct.pack("<d", float(v))
    if isinstance(v, str):
        b = v.encode("utf-8") + b"\x00"
        return b"\x02" + k + struct.pack("<i", len(b)) + b
    if isinstance(v, (bytes, bytearray, memoryview)):
        return b"\x05" + k + struct.pack("<i", len(v)) + b"\x00" + bytes(v)
    if isinstance(v, dict):
        return b"\x03" + k + dumps(v)
    if isinstance(v, (list, tuple)):
        return b"\x04" + k + dumps({str(i): x for i, x in enumerate(v)})
    raise TypeError(f"no BSON encoding for {type(v).__name__}")


def dumps(doc: dict) -> bytes:
    """One BSON document. Keys keep their insertion order."""
    body = b"".join(_elem(str(k), v) for k, v in doc.items())
    return struct.pack("<i", len(body) + 5) + body + b"\x00"


def _parse_doc(b: memoryview, at: int, as_list: bool):
    (n,) = struct.unpack_from("<i", b, at)
    if n < 5 or at + n > len(b) or b[at + n - 1] != 0:
        raise ValueError("malformed BSON document")
    end, at = at + n - 1, at + 4
    out = [] if as_list else {}
    while at < end:
        t = b[at]
        z, w = -1, 64          # the key's terminator, looked for in growing windows: O(key length), not a copy of the document's remainder per element
        while z < 0 and at + 1 < end:
            z = bytes(b[at + 1:min(end, at + 1 + w)]).find(b"\x00")
            if at + 1 + w >= end:
                break
            w *= 8
        if z < 0:
            raise ValueError("unterminated BSON key")
        key, at = bytes(b[at + 1:at + 1 + z]).decode("utf-8"), at + 2 + z
        if t == 0x01:
            v, at = struct.unpack_from("<d", b, at)[0], at + 8
        elif t == 0x02:
            (m,) = struct.unpack_from("<i", b, at)
            v, at = bytes(b[at + 4:at + 3 + m]).decode("utf-8"), at + 4 + m
        elif t in (0x03, 0x04):
            v, at = _parse_doc(b, at, t == 0x04)
        elif t == 0x05:
            (m,) = struct.unpack_from("<i", b, at)
            v, at = bytes(b[at + 5:at + 5 + m]), at + 5 + m
        elif t == 0x08:
            v, at = b[at] != 0, at + 1
        elif t == 0x0A:
            v = None
        elif t == 0x10:
            v, at = struct.unpack_from("<i", b, at)[0], at + 4
        elif t == 0x12:
            v, at = struct.unpack_from("<q", b, at)[0], at + 8
        else:
            raise ValueError(f"BSON element type 0x{t:02x} is not used by this container")
        if as_list:
            out.append(v)                                                 # (array keys are positional; their text is ignored, as BSON.jl does)
        else:
            out[key] = v
    if at != end:
        raise ValueError("malformed BSON document")
    return out, end + 1


def loads(data: bytes) -> dict:
    doc, at = _parse_doc(memoryview(data), 0, False)
    if at != len(data):
        raise ValueError("trailing bytes after the BSON document")
    return doc


# ---- Julia values <-> documents (BSON.jl's lowering) ---------------------------------------------------------------------
def _datatype(name, params=()):
    return {"tag": "datatype", "name": ["Core", name], "params": list(params)}


def _array_type(a: np.ndarray):
    return _datatype("Array", [_datatype(_JL[a.dtype]), a.ndim])


def lower(x):
    """numpy array → Array{T,N} (column-major bytes); list → Vector of the lowered elements; tuple → Tuple; scalars and strings as
    they are; dict → document of lowered values."""
    if isinstance(x, np.ndarray):
        if x.dtype not in _JL:
            raise TypeError(f"no Julia bits type for dtype {x.dtype}")
        return {"tag": "array", "type": _datatype(_JL[x.dtype]), "size": [int(d) for d in x.shape],
                "data": np.asfortranarray(x).tobytes(order="F")}
    if isinstance(x, tuple):
        return {"tag": "tuple", "data": [lower(v) for v in x]}
    if isinstance(x, list):
        el = [lower(v) for v in x]
        first = x[0] if x else None
        if isinstance(first, np.ndarray):
            ty = _array_type(first)
        elif isinstance(first, list) and first and isinstance(first[0], np.ndarray):
            ty = _datatype("Array", [_array_type(first[0]), 1])
        else:
            ty = _datatype("Any")
        return {"tag": "array", "type": ty, "size": [len(x)], "data": el}
    if isinstance(x, dict):
        return {str(k): lower(v) for k, v in x.items()}
    return x


def raise_(d):
    """The inverse of `lower`."""
    if isinstance(d, list):
        return [raise_(v) for v in d]
    if not isinstance(d, dict):
        return d
    tag = d.get("tag")
    if tag == "tuple":
        return tuple(raise_(v) for v in d["data"])
    if tag == "array":
        ty, size, data = d["type"], [int(s) for s in d["size"]], d["data"]
        if isinstance(data, (bytes, bytearray)):
            name = ty["name"][-1]
            if name not in _NP:
                raise TypeError(f"array of Julia type {name} is not supported")
            a = np.frombuffer(data, dtype=_NP[name])
            if a.size != int(np.prod(size, dtype=np.int64)):
                raise ValueError("array data does not match its size")
            return a.reshape(size, order="F").copy()
        if len(size) != 1 or size[0] != len(data):
            raise ValueError("only vectors of non-bits elements are supported")
        return [raise_(v) for v in data]
    if tag is not None:
        raise TypeError(f"BSON.jl tag {tag!r} is not supported")
    return {k: raise_(v) for k, v in d.items()}


def save(path: str, **named):
    """`@save path a b …` — one document with a key per variable."""
    with open(path, "wb") as f:
        f.write(dumps({k: lower(v) for k, v in named.items()}))
    return path


def load(path: str) -> dict:
    with open(path, "rb") as f:
        return raise_(loads(f.read()))


# ---- the reference's data tuple ---------------------------------------------------------------------------------------------
def dataset_to_julia(latent_data, u0s, ps, high_dim_data):
    """Dense arrays (latent [2, T, n], u0s [2, n], ps [P, n], frames [h, w, T, n]) → the reference's nesting."""
    latent_data, u0s, ps, high = (np.asarray(a, dtype=np.float32) for a in (latent_data, u0s, ps, high_dim_data))
    n = latent_data.shape[-1]
    if not (u0s.shape[-1] == ps.shape[-1] == high.shape[-1] == n and high.shape[2] == latent_data.shape[1]):
        raise ValueError("dataset arrays disagree on the number of trajectories or save points")
    return ([np.ascontiguousarray(latent_data[:, :, i]) for i in range(n)],
            [np.ascontiguousarray(u0s[:, i]) for i in range(n)],
            [np.ascontiguousarray(ps[:, i]).reshape(-1, 1) for i in range(n)],          # rand_uniform(range, size) is size × 1
            [[np.ascontiguousarray(high[:, :, t, i]) for t in range(high.shape[2])] for i in range(n)])


def dataset_from_julia(data):
    latent_data, u0s, ps, high = data
    return (np.stack(latent_data, axis=-1), np.stack(u0s, axis=-1), np.stack([p.reshape(-1) for p in ps], axis=-1),
            np.stack([np.stack(fr, axis=-1) for fr in high], axis=-1))


def dump_dataset(path: str, latent_data, u0s, ps, high_dim_data) -> str:
    return save(path, data=dataset_to_julia(latent_data, u0s, ps, high_dim_data))


def load_dataset(path: str):
    doc = load(path)
    if "data" not in doc or not isinstance(doc["data"], tuple) or len(doc["data"]) != 4:
        raise KeyError(f"{path}: no 4-tuple `data` in this BSON file")
    return dataset_from_julia(doc["data"])
