"""Readers / writers for the reference's on-disk scan formats (harness side of the path;
SURVEY.md 8(f) rank 4).

* `scans/2d/NNN.txt` -- one `x y` per line, single space (examples/scan2d.rs:18-34).
* `scans/3d/scans.hdf5` -- one HDF5 dataset per LiDAR packet, each an f64 array of 24 x 16 x 3 values
  that the example reshapes to 384 x 3 points (examples/scan3d.rs:9,21-23,45-49), read in the file's
  dataset order (`file.datasets()`, :36-37).  libhdf5 / h5py do not exist in this image and the
  recording itself is absent from the reference mount, so the packet stream is carried by a small
  self-describing container with the SAME logical layout (named datasets, one packet each, rank-3
  f64 little-endian), `PacketFile`; `convert` instructions for a real recording are in INTEGRATION.md.

Container layout (all little-endian):
    0   8   magic  b"ICPPKT01"
    8   4   u32    n_datasets
    12  4   u32    rank (3)
    16  12  u32[3] dims of every dataset (24, 16, 3)
    28  4   u32    dtype code (1 = IEEE f64)
    32  32  reserved (zero)
    64  n_datasets x 48: name[32] (NUL padded, the HDF5 dataset name), u64 offset, u64 nbytes
    ... dataset payloads, 64-byte aligned, C order
"""
import struct

import numpy as np

N_POINTS_IN_PACKET = 24 * 16  # examples/scan3d.rs:9
MAGIC = b"ICPPKT01"
_HEADER = struct.Struct("<8sII3II32x")
_ENTRY = struct.Struct("<32sQQ")
_DTYPE_F64 = 1


def load_scan2d(path):
    """examples/scan2d.rs:18-34: one `x y` per line, split on a single space."""
    pts = []
    with open(path) as f:
        for line in f:
            s = line.rstrip("\n")
            xy = s.split(" ")
            pts.append((float(xy[0]), float(xy[1])))
    return np.array(pts, dtype=np.float64).reshape(-1, 2)


def save_scan2d(path, points):
    """The writer twin of load_scan2d: `repr` of a double round-trips exactly, like the full-precision
    decimals of the reference's scans."""
    p = np.asarray(points, dtype=np.float64).reshape(-1, 2)
    with open(path, "w") as f:
        for x, y in p:
            f.write(f"{float(x)!r} {float(y)!r}\n")


def write_packets(path, packets, dims=(24, 16, 3), names=None):
    """Write `packets` (n, 384, 3) -- or (n, *dims) -- as one dataset per packet."""
    arr = np.ascontiguousarray(packets, dtype="<f8")
    n = arr.shape[0]
    per = int(np.prod(dims))
    if arr.size != n * per:
        raise ValueError(f"packets of {arr.size // max(n, 1)} values do not fill datasets of dims {dims}")
    arr = arr.reshape(n, per)
    if names is None:
        names = [f"{k:06d}" for k in range(n)]  # HDF5 iterates datasets by name: zero-padded = packet order
    if len(names) != n or len(set(names)) != n:
        raise ValueError("one unique name per dataset")
    table = _HEADER.size + n * _ENTRY.size
    first = (table + 63) // 64 * 64
    stride = (per * 8 + 63) // 64 * 64
    with open(path, "wb") as f:
        f.write(_HEADER.pack(MAGIC, n, len(dims), *dims, _DTYPE_F64))
        for k, name in enumerate(names):
            b = name.encode()
            if len(b) > 31:
                raise ValueError("dataset names are at most 31 bytes")
            f.write(_ENTRY.pack(b, first + k * stride, per * 8))
        f.write(b"\0" * (first - table))
        pad = b"\0" * (stride - per * 8)
        for k in range(n):
            f.write(arr[k].tobytes())
            f.write(pad)


class PacketFile:
    """`Scan` of examples/scan3d.rs:17-61 over the container: `size()`, `get(index)` (384 x 3 points of
    one packet), `get_range(start, end)` (concatenated)."""

    def __init__(self, path):
        self.path = path
        self._mm = np.memmap(path, dtype=np.uint8, mode="r")
        if self._mm.size < _HEADER.size:
            raise ValueError(f"{path}: not a packet container (too short)")
        magic, n, rank, d0, d1, d2, dtype = _HEADER.unpack(self._mm[:_HEADER.size].tobytes())
        if magic != MAGIC:
            raise ValueError(f"{path}: not a packet container (bad magic)")
        if rank != 3 or dtype != _DTYPE_F64:
            raise ValueError(f"{path}: rank {rank} / dtype {dtype} not supported (rank-3 f64 datasets)")
        self.dims = (d0, d1, d2)
        if d0 * d1 * d2 != N_POINTS_IN_PACKET * 3:
            # the reference's reshape((384, 3)).unwrap() panics on anything else (scan3d.rs:21-23)
            raise ValueError(f"{path}: datasets of {self.dims} do not reshape to ({N_POINTS_IN_PACKET}, 3)")
        self.names, self._off = [], []
        size = self._mm.size
        if _HEADER.size + n * _ENTRY.size > size:
            raise ValueError(f"{path}: the table of {n} datasets does not fit the file")
        for k in range(n):
            lo = _HEADER.size + k * _ENTRY.size
            name, off, nbytes = _ENTRY.unpack(self._mm[lo:lo + _ENTRY.size].tobytes())
            if nbytes != N_POINTS_IN_PACKET * 24 or off + nbytes > size:
                raise ValueError(f"{path}: dataset {k} is truncated or mis-sized")
            self.names.append(name.rstrip(b"\0").decode())
            self._off.append(off)

    def size(self):
        return len(self._off)

    def __len__(self):
        return len(self._off)

    def get(self, index):
        off = self._off[index]
        raw = self._mm[off:off + N_POINTS_IN_PACKET * 24]
        return np.frombuffer(raw.tobytes(), dtype="<f8").reshape(N_POINTS_IN_PACKET, 3).astype(np.float64)

    def get_range(self, start, end):
        if end <= start:
            return np.zeros((0, 3))
        return np.concatenate([self.get(i) for i in range(start, end)])

    def as_array(self):
        """(n_packets, 384, 3)"""
        return np.stack([self.get(i) for i in range(self.size())]) if self.size() else np.zeros((0, N_POINTS_IN_PACKET, 3))

    def close(self):
        self._mm = None


def is_packet_file(path):
    try:
        with open(path, "rb") as f:
            return f.read(8) == MAGIC
    except OSError:
        return False


def file_size_for(n_packets):
    """bytes a container of n packets occupies (header + table + 64-byte aligned payloads)"""
    table = _HEADER.size + n_packets * _ENTRY.size
    return (table + 63) // 64 * 64 + n_packets * ((N_POINTS_IN_PACKET * 24 + 63) // 64 * 64)


__all__ = ["load_scan2d", "save_scan2d", "write_packets", "PacketFile", "is_packet_file", "N_POINTS_IN_PACKET",
           "file_size_for"]
