"""Minimal PLY reader / writer and the CLI's column handling (next row 8f-3).

Reference behaviour: ``pointstowood/src/io.py:11-83`` (``read_ply`` / ``write_ply``: vertex-only PLY, ascii or binary,
output = binary little-endian with x, y, z and every other column as float64, red/green/blue as int) and
``pointstowood/predict.py:36-52`` (``preprocess_point_cloud_data``).  numpy only; a point cloud is an ordered
``dict`` column name -> 1-D array.
"""
from __future__ import annotations

import numpy as np

_PLY_TYPES = {
    "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
    "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8",
}


def read_ply(path):
    """-> dict name -> array (file order).  Vertex element only; a face element is refused like the reference does."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, n, props, in_vertex = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            tok = line.decode("latin-1").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n = int(tok[2])
                elif tok[1] == "face" and int(tok[2]) > 0:
                    raise ValueError(".ply appears to be a mesh")
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("list properties are not supported on vertices")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt is None or n is None:
            raise ValueError(f"{path}: PLY header without format / vertex element")
        if fmt == "ascii":
            arr = np.loadtxt(f, dtype=np.float64, max_rows=n, ndmin=2)
            if arr.shape[0] != n or arr.shape[1] < len(props):
                raise ValueError(f"{path}: expected {n} x {len(props)} values")
            return {name: arr[:, i].astype(t) for i, (name, t) in enumerate(props)}
        order = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(name, order + t) for name, t in props])
        rec = np.fromfile(f, dtype=dt, count=n)
        if rec.shape[0] != n:
            raise ValueError(f"{path}: truncated PLY body ({rec.shape[0]} of {n} vertices)")
        return {name: np.ascontiguousarray(rec[name]).astype(rec[name].dtype.newbyteorder("=")) for name, _ in props}


def write_ply(path, columns, comments=()):
    """Binary little-endian PLY, byte for byte what the reference's ``write_ply`` emits for the same columns
    (io.py:49-83): x, y, z float64, red/green/blue ``int`` right after them when all three are present, every other
    column float64 in input order; the header carries the reference's fixed comment / obj_info lines so that a file
    written here is indistinguishable from one written there (checked against a reference-written fixture)."""
    names = ["x", "y", "z"] + (["red", "green", "blue"] if "red" in columns else [])
    n = len(columns["x"])
    data = {c: np.asarray(columns[c]).reshape(n) for c in names}
    for c in columns:
        if c in data:
            continue
        try:   # the reference drops, silently, every extra column that does not convert to float64 (io.py:72-78)
            data[c] = np.asarray(columns[c]).reshape(n).astype(np.float64)
            names.append(c)
        except (TypeError, ValueError):
            pass
    dt = np.dtype([(c, "<i4" if c in ("red", "green", "blue") and "red" in columns else "<f8") for c in names])
    rec = np.empty(n, dtype=dt)
    for c in names:
        rec[c] = data[c]
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\ncomment Author: Phil Wilkes\n")
        for c in comments:
            f.write(f"comment {c}\n".encode())
        f.write(b"obj_info generated with pcd2ply.py\n")
        f.write(f"element vertex {n}\n".encode())
        for c in names:
            f.write(f"property {'int' if dt[c] == np.dtype('<i4') else 'float64'} {c}\n".encode())
        f.write(b"end_header\n")
        rec.tofile(f)


def prepare_columns(columns):
    """``preprocess_point_cloud_data`` (predict.py:36-52): lower-case names, drop label / pwood / pleaf, strip
    ``scalar_``, refl | intensity -> reflectance, add a zero reflectance if there is none, move reflectance to column
    3.  -> (columns, headers, had_reflectance) where ``headers`` are the input's extra columns (the ones echoed to the
    output)."""
    drop = ("label", "pwood", "pleaf")
    out = {}
    for name, v in columns.items():
        name = name.lower()
        if name in drop:
            continue
        name = name.replace("scalar_", "")
        name = {"refl": "reflectance", "intensity": "reflectance"}.get(name, name)
        out[name] = v
    names = list(out)
    headers = [h for h in names[3:] if h not in drop]
    had = "reflectance" in out
    if not had:
        out["reflectance"] = np.zeros(len(out[names[0]]))
    names = list(out)
    names.insert(3, names.pop(names.index("reflectance")))
    return {k: out[k] for k in names}, headers, had
