"""proto2 wire format for the "bags" the model surface exchanges (suff-stats / hyper-parameters as bytes,
microscopes/models.pyx:68-94): a dozen lines of varint / fixed32 / length-delimited fields, no protobuf runtime.

The in-tree messages follow microscopes/io/schema.proto:3-46 and are pinned byte for byte against Google's protobuf
runtime (tests/golden/wire.json).  The six `distributions` messages (bb, bnb, gp, nich, dd, niw) follow that absent
library's published schema as restated in include/microscopes_amd/wire.hpp; their bytes are unpinned.
"""
import struct


def _varint(v):
    v &= (1 << 64) - 1                      # negative int32 / int64 travel as 10-byte two's complement varints
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def put_varint(field, v):
    return _varint(field << 3 | 0) + _varint(int(v))


def put_float(field, v):
    return _varint(field << 3 | 5) + struct.pack("<f", float(v))


def put_bytes(field, b):
    b = bytes(b)
    return _varint(field << 3 | 2) + _varint(len(b)) + b


def parse(buf):
    """-> [(field number, wire type, value)]: int for varints, float for fixed32, bytes for length-delimited"""
    buf = bytes(buf)
    i, out = 0, []

    def rd():
        nonlocal i
        v, shift = 0, 0
        while True:
            if i >= len(buf) or shift > 63:
                raise ValueError("malformed varint")
            b = buf[i]
            i += 1
            v |= (b & 0x7F) << shift
            if not b & 0x80:
                return v
            shift += 7
    while i < len(buf):
        key = rd()
        num, wt = key >> 3, key & 7
        if wt == 0:
            out.append((num, wt, rd()))
        elif wt == 5:
            if i + 4 > len(buf):
                raise ValueError("truncated fixed32")
            out.append((num, wt, struct.unpack_from("<f", buf, i)[0]))
            i += 4
        elif wt == 2:
            n = rd()
            if n > len(buf) - i:
                raise ValueError("truncated bytes field")
            out.append((num, wt, buf[i:i + n]))
            i += n
        else:
            raise ValueError("unsupported wire type %d" % wt)
    return out


def _floats(fields, num):
    """repeated float: one per field, or packed in a length-delimited field"""
    out = []
    for n, wt, v in fields:
        if n != num:
            continue
        if wt == 5:
            out.append(v)
        elif wt == 2:
            out.extend(struct.unpack("<%df" % (len(v) // 4), v))
    return out


def _varints(fields, num):
    out = []
    for n, wt, v in fields:
        if n != num:
            continue
        if wt == 0:
            out.append(v)
        elif wt == 2:
            out.extend(_unpack_varints(v))
    return out


def _unpack_varints(b):
    i, out = 0, []
    while i < len(b):
        v, shift = 0, 0
        while True:
            c = b[i]
            i += 1
            v |= (c & 0x7F) << shift
            if not c & 0x80:
                break
            shift += 7
        out.append(v)
    return out


# message layouts: name -> [(key, field number, kind)], kind in f (float), u (uint varint), F / U (repeated)
SCHEMA = {
    "crp": [("alpha", 1, "f")],                                                     # schema.proto:3-5
    "bbnc.shared": [("alpha", 1, "f"), ("beta", 2, "f")],                           # :8-11
    "bbnc.group": [("p", 1, "f"), ("heads", 2, "u"), ("tails", 3, "u")],            # :13-17
    "dm.shared": [("alphas", 1, "F")],                                              # :21-23
    "dm.group": [("counts", 1, "U"), ("ratio", 2, "f")],                            # :25-28
    # the absent library's messages (unpinned)
    "bb.shared": [("alpha", 1, "f"), ("beta", 2, "f")],
    "bb.group": [("heads", 1, "u"), ("tails", 2, "u")],
    "bnb.shared": [("alpha", 1, "f"), ("beta", 2, "f"), ("r", 3, "u")],
    "bnb.group": [("count", 1, "u"), ("sum", 2, "u")],
    "gp.shared": [("alpha", 1, "f"), ("inv_beta", 2, "f")],
    "gp.group": [("count", 1, "u"), ("sum", 2, "u"), ("log_prod", 3, "f")],
    "nich.shared": [("mu", 1, "f"), ("kappa", 2, "f"), ("sigmasq", 3, "f"), ("nu", 4, "f")],
    "nich.group": [("count", 1, "u"), ("mean", 2, "f"), ("count_times_variance", 3, "f")],
    "dd.shared": [("alphas", 1, "F")],
    "dd.group": [("counts", 1, "U")],
    "niw.shared": [("mu", 1, "F"), ("kappa", 2, "f"), ("psi", 3, "F"), ("nu", 4, "f")],
    "niw.group": [("count", 1, "u"), ("sum_x", 2, "F"), ("sum_xxT", 3, "F")],
}


def dumps(message, d):
    out = b""
    for key, num, kind in SCHEMA[message]:
        v = d[key]
        if kind == "f":
            out += put_float(num, v)
        elif kind == "u":
            out += put_varint(num, v)
        elif kind == "F":
            for x in _flat(v):
                out += put_float(num, x)
        else:
            for x in _flat(v):
                out += put_varint(num, x)
    return out


def loads(message, raw):
    fields = parse(raw)
    d = {}
    for key, num, kind in SCHEMA[message]:
        if kind == "F":
            d[key] = _floats(fields, num)
        elif kind == "U":
            d[key] = [int(x) for x in _varints(fields, num)]
        else:
            hit = [v for n, _, v in fields if n == num]
            if not hit:
                raise ValueError("%s: required field %s missing" % (message, key))
            d[key] = float(hit[-1]) if kind == "f" else int(hit[-1])
    return d


def _flat(v):
    try:
        import numpy as np
        return [x for x in np.asarray(v).ravel().tolist()]
    except Exception:
        return list(v)


def group_manager_dumps(alpha, assignments, groups):
    """io::GroupManager (schema.proto:31-40) as group_manager::serialize writes it (group_manager.hpp:285-298):
    groups = [(id, data bytes)] in ascending id order."""
    out = put_float(1, alpha)
    for a in assignments:
        out += put_varint(2, int(a))
    for gid, data in groups:
        out += put_bytes(3, put_varint(1, gid) + put_bytes(2, data))
    return out


def group_manager_loads(raw):
    alpha, assignments, groups = None, [], []
    for num, wt, v in parse(raw):
        if num == 1:
            alpha = v
        elif num == 2:
            vals = [v] if wt == 0 else _unpack_varints(v)
            assignments.extend(x - (1 << 64) if x >> 63 else x for x in vals)
        elif num == 3:
            gid, data = None, b""
            for n2, _, v2 in parse(v):
                if n2 == 1:
                    gid = v2
                elif n2 == 2:
                    data = v2
            groups.append((gid, data))
    return alpha, assignments, groups
