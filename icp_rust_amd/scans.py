"""Readers for the reference's on-disk scan formats (harness side of the path)."""
import numpy as np


def load_scan2d(path):
    """examples/scan2d.rs:18-34: one `x y` per line, split on a single space."""
    pts = []
    with open(path) as f:
        for line in f:
            s = line.rstrip("\n")
            xy = s.split(" ")
            pts.append((float(xy[0]), float(xy[1])))
    return np.array(pts, dtype=np.float64).reshape(-1, 2)
