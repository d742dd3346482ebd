"""PLY wire format of a GaussianModel: interchange with the reference's checkpoints / viewers.

`GaussianModel.save_ply` / `load_ply` (gs_renderer.py:713-744, 769-852) go through the `plyfile` package; the format
itself is: one `vertex` element, every property `float` (f4), binary little endian, in the order
    x y z  nx ny nz  f_dc_0..2  f_rest_0..(3(K-1)-1)  opacity  scale_0..2  rot_0..3
with the SH features flattened CHANNEL-major (`_features_dc.transpose(1, 2).flatten(1)`: all coefficients of R, then G,
then B) although the tensors are coefficient-major [P, K-1, 3]; normals are zeros; opacity / scale / rot hold the RAW
(pre-activation) values. This module reads and writes that format with numpy only (host-side I/O, no GPU work).
`load_ply` also accepts ascii and big-endian files and properties in any order / of other scalar types.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import numpy as np

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
              "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
              "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def attribute_names(n_rest_coeffs: int) -> List[str]:
    """Property order of construct_list_of_attributes (gs_renderer.py:713-726) for K - 1 = n_rest_coeffs."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(3)]
    names += [f"f_rest_{i}" for i in range(3 * n_rest_coeffs)]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(3)]
    names += [f"rot_{i}" for i in range(4)]
    return names


def pack_attributes(xyz, features_dc, features_rest, opacity, scaling, rotation) -> np.ndarray:
    """[P, 17 + 3K] float32 matrix in file order from the leaf layouts ([P,3], [P,1,3], [P,K-1,3], [P,1], [P,3], [P,4])."""
    f32 = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    xyz = f32(xyz)
    P = xyz.shape[0]
    f_dc = f32(features_dc).reshape(P, -1, 3).transpose(0, 2, 1).reshape(P, -1)
    f_rest = f32(features_rest).reshape(P, -1, 3).transpose(0, 2, 1).reshape(P, -1)
    return np.concatenate((xyz, np.zeros_like(xyz), f_dc, f_rest, f32(opacity).reshape(P, 1), f32(scaling).reshape(P, 3),
                           f32(rotation).reshape(P, 4)), axis=1)


def save_ply(path: str, xyz, features_dc, features_rest, opacity, scaling, rotation) -> None:
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    attrs = pack_attributes(xyz, features_dc, features_rest, opacity, scaling, rotation)
    names = attribute_names((attrs.shape[1] - 17) // 3)
    assert len(names) == attrs.shape[1]
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {attrs.shape[0]}"]
    header += [f"property float {n}" for n in names]
    header.append("end_header")
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(attrs.astype("<f4", copy=False).tobytes())


def _read_vertex_table(path: str) -> Dict[str, np.ndarray]:
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elements, cur = None, [], None
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                cur = dict(name=tok[1], count=int(tok[2]), props=[])
                elements.append(cur)
            elif tok[0] == "property":
                if tok[1] == "list":
                    if cur["name"] == "vertex":
                        raise ValueError(f"{path}: list property on the vertex element")
                    cur["props"].append((tok[4], None))
                else:
                    if tok[1] not in _PLY_TYPES:
                        raise ValueError(f"{path}: unknown property type {tok[1]}")
                    cur["props"].append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if not elements or elements[0]["name"] != "vertex":
            raise ValueError(f"{path}: the first element must be `vertex` (it is what load_ply reads, :770)")
        v = elements[0]
        if fmt == "ascii":
            rows = [f.readline().split() for _ in range(v["count"])]
            tab = np.array(rows, dtype=np.float64).reshape(v["count"], len(v["props"]))
            return {n: tab[:, k] for k, (n, _) in enumerate(v["props"])}
        if fmt not in ("binary_little_endian", "binary_big_endian"):
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
        bo = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, bo + t) for n, t in v["props"]])
        raw = f.read(dt.itemsize * v["count"])
        if len(raw) != dt.itemsize * v["count"]:
            raise ValueError(f"{path}: truncated vertex data")
        rec = np.frombuffer(raw, dtype=dt, count=v["count"])
        return {n: rec[n] for n, _ in v["props"]}


def load_ply(path: str, max_sh_degree: Optional[int] = None) -> Dict[str, np.ndarray]:
    """Leaf tensors as GaussianModel.load_ply builds them (gs_renderer.py:769-852): float32 numpy arrays
    xyz [P,3], features_dc [P,1,3], features_rest [P,K-1,3], opacity [P,1], scaling [P,3], rotation [P,4].
    max_sh_degree: K = (max_sh_degree + 1)^2 as the loading model prescribes (extra f_rest_* properties are dropped,
    missing ones stay zero, like the reference's loop :790-794); None = take K from the file."""
    t = _read_vertex_table(path)
    P = len(t["x"])
    col = lambda n: np.asarray(t[n], dtype=np.float32)
    xyz = np.stack((col("x"), col("y"), col("z")), axis=1)
    opacity = col("opacity")[:, None]
    features_dc = np.stack((col("f_dc_0"), col("f_dc_1"), col("f_dc_2")), axis=1)[:, None, :]      # [P,1,3]
    extra = [n for n in t if n.startswith("f_rest_")]          # file order, as the reference enumerates them
    n_file = len(extra)
    n_want = n_file if max_sh_degree is None else 3 * (max_sh_degree + 1) ** 2 - 3
    if n_want % 3:
        raise ValueError(f"{path}: {n_file} f_rest_* properties is not a multiple of 3")
    flat = np.zeros((P, n_want), dtype=np.float32)
    for idx, name in enumerate(extra[:n_want]):
        flat[:, idx] = col(name)
    features_rest = np.ascontiguousarray(flat.reshape(P, 3, n_want // 3).transpose(0, 2, 1))      # [P,K-1,3]
    scale_names = [n for n in t if n.startswith("scale_")]
    rot_names = [n for n in t if n.startswith("rot")]
    scaling = np.stack([col(n) for n in scale_names], axis=1)
    rotation = np.stack([col(n) for n in rot_names], axis=1)
    return dict(xyz=xyz, features_dc=np.ascontiguousarray(features_dc), features_rest=features_rest, opacity=opacity,
                scaling=scaling, rotation=rotation)


def load_model(path: str, device, max_sh_degree: Optional[int] = None, requires_grad: bool = True):
    """A render_api.GaussianParams on `device` from a PLY (active_sh_degree = the file's full degree, :852)."""
    import torch
    from .render_api import GaussianParams
    d = load_ply(path, max_sh_degree)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=device).requires_grad_(requires_grad)
    K = 1 + d["features_rest"].shape[1]
    deg = {1: 0, 4: 1, 9: 2, 16: 3}.get(K)
    if deg is None:
        raise ValueError(f"{path}: {K} SH coefficients per channel is not (degree+1)^2 for degree <= 3")
    return GaussianParams(t(d["xyz"]), t(d["scaling"]), t(d["rotation"]), t(d["opacity"]), t(d["features_dc"]),
                          t(d["features_rest"]), deg)
