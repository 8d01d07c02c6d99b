"""MAT-file Level 4 helpers for the tests (numpy only): the wire format of the
reference's Input.mat / Output.mat (DTSource/DTMatlabDataFile.cpp:28-35,456-492):
20-byte header {type, mrows, ncols, imagf, namelen} + NUL-terminated name +
column-major data."""
import struct

import numpy as np

_DT = {0: "f8", 1: "f4", 2: "i4", 3: "i2", 4: "u2", 5: "u1"}


def write(path, variables, big_endian=False, ptype=0):
    """variables: list of (name, 2-D array).  ptype: element type code P."""
    e = ">" if big_endian else "<"
    with open(path, "wb") as fh:
        for name, a in variables:
            a = np.atleast_2d(np.asarray(a))
            if a.ndim == 2 and a.shape[0] == 1 and np.asarray(variables[0][1]).ndim == 1:
                pass
            m, n = a.shape
            fh.write(struct.pack(e + "5i", (1000 if big_endian else 0) + 10 * ptype, m, n, 0, len(name) + 1))
            fh.write(name.encode() + b"\0")
            fh.write(np.asfortranarray(a).astype(e + _DT[ptype]).tobytes(order="F"))


def read(path):
    """{name: 2-D float64 array} of every numeric variable in the file."""
    out = {}
    data = open(path, "rb").read()
    pos = 0
    while pos + 20 <= len(data):
        t, m, n, imagf, namelen = struct.unpack_from("<5i", data, pos)
        e = "<"
        if t < 0 or t > 52:
            t, m, n, imagf, namelen = struct.unpack_from(">5i", data, pos)
            e = ">"
        t %= 1000
        name = data[pos + 20:pos + 20 + namelen - 1].decode()
        dt = np.dtype(e + _DT[(t % 100) // 10])
        cnt = m * n
        start = pos + 20 + namelen
        out[name] = np.frombuffer(data, dtype=dt, count=cnt, offset=start).astype(np.float64).reshape((m, n), order="F")
        pos = start + cnt * dt.itemsize
    return out
