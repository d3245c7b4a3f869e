"""Generates tests/golden/self_golden.json: poses the CPU oracle (reference summation order,
kd-tree NN) produces for (a) the scan2d frame loop over the committed scans 001..040 and
(b) the reference's own 21-point "L" cases (src/lib.rs:509-595).  SELF-golden: there is no Rust
toolchain here to produce them with the reference binary; they freeze the oracle so that any
later change to it (or to libm) shows up as a diff.  Run from the repo root:
    python tests/golden/make_self_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_ffi as O  # noqa: E402
from icp_rust_amd.scans import load_scan2d  # noqa: E402


def main():
    out = {"note": "self-golden (oracle-generated, reference summation order); poses as "
                   "[r00, r10, r01, r11, tx, ty]; floats as repr() strings round-trip exactly"}
    src = load_scan2d(os.path.join(HERE, "scans2d", "001.txt"))
    T = O.transform_identity()
    frames = []
    for k in range(2, 41):  # examples/scan2d.rs:62-90: warm start from the previous frame
        dst = load_scan2d(os.path.join(HERE, "scans2d", f"{k:03d}.txt"))
        rc, T, idx, inner = O.icp_estimate(2, dst, src, T, 20, use_kdtree=True)
        assert rc == O.OK
        frames.append({"frame": k, "pose": [repr(float(x)) for x in T.as_array()],
                       "inner_iters": [int(x) for x in inner],
                       "idx_checksum": int(np.bitwise_xor.reduce(idx.astype(np.uint64) * np.arange(1, len(idx) + 1, dtype=np.uint64)))})
    out["scan2d"] = frames
    L = np.array([[0.0, v] for v in (0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0)] +
                 [[v, 0.0] for v in (0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0)])
    Tt = O.transform_new(np.array([0.01, 0.01, -0.02]))
    Ti = O.transform_mul(O.transform_new(np.array([0.05, 0.010, 0.010])), Tt)
    dst2 = O.transform_apply_many(Tt, L)
    rc, T2, _, in2 = O.icp_estimate(2, dst2, L, Ti, 20, use_kdtree=True)
    src3 = np.concatenate([L, np.array([[2.0]] * 11 + [[1.0]] * 10)], axis=1)
    dst3 = np.array([O.transform_xy(Tt, p) for p in src3])
    rc, T3, _, in3 = O.icp_estimate(3, dst3, src3, Ti, 20, use_kdtree=True)
    out["l_shape_2d"] = {"pose": [repr(float(x)) for x in T2.as_array()], "inner_iters": [int(x) for x in in2]}
    out["l_shape_3d"] = {"pose": [repr(float(x)) for x in T3.as_array()], "inner_iters": [int(x) for x in in3]}
    with open(os.path.join(HERE, "self_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", os.path.join(HERE, "self_golden.json"))


if __name__ == "__main__":
    main()
