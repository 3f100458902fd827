#!/usr/bin/env python3
"""Generates tests/golden/femur_mesh.npz: the triangulations of the reference's femur STL pair
(/root/reference/examples/data/femur -- data, not source) in the vertex numbering of tests/golden/inputs.npz
(first-occurrence de-duplication of the binary STL's triangle corners, see make_golden.py).  Run from the repo root IN
THE BUILD CONTAINER:   python tests/golden/make_mesh_fixture.py
"""
import os
import struct

import numpy as np

DATA = "/root/reference/examples/data"
OUT = os.path.dirname(os.path.abspath(__file__))


def read_binary_stl(path):
    raw = open(path, "rb").read()
    n = struct.unpack("<I", raw[80:84])[0]
    tri = np.frombuffer(raw, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), count=n, offset=84)
    corners = tri["v"].reshape(-1, 3)
    seen, order, ids = {}, [], np.empty(corners.shape[0], dtype=np.int32)
    for k, c in enumerate(corners):
        key = c.tobytes()
        if key not in seen:
            seen[key] = len(order)
            order.append(c)
        ids[k] = seen[key]
    return np.asarray(order, dtype=np.float32), ids.reshape(-1, 3)


def main():
    inputs = np.load(f"{OUT}/inputs.npz")
    v, c = read_binary_stl(f"{DATA}/femur/femur.stl")
    vt, ct = read_binary_stl(f"{DATA}/femur/femur_target.stl")
    assert np.array_equal(v, inputs["femur"]) and np.array_equal(vt, inputs["femur_target"])
    np.savez_compressed(f"{OUT}/femur_mesh.npz", femur_cells=c, femur_target_cells=ct)
    print("cells:", c.shape, ct.shape)


if __name__ == "__main__":
    main()
