"""A minimal FlatBuffers builder that lays buffers out the way the official builders (flatc-generated Rust/C++/Python code) do:
back to front, children before parents, the vtable in front of its table (positive soffset), shared padding rules.  The product's
own .zkif writer emits a forward layout; this one exists so that tests can feed the reader the layout real zkInterface producers
(the Haskell/CirC front ends of the reference, through the `zkinterface` crate) write.  Test infrastructure only."""
import struct


class Builder:
    def __init__(self):
        self.b = bytearray()          # bytes of the finished buffer's TAIL; we only ever prepend
        self.minalign = 1
        self.vtables = {}

    def offset(self):
        return len(self.b)

    def _pad(self, n):
        self.b[0:0] = bytes(n)

    def prep(self, size, additional):
        self.minalign = max(self.minalign, size)
        align = (~(len(self.b) + additional) + 1) & (size - 1)
        self._pad(align)

    def _prepend(self, fmt, v):
        self.b[0:0] = struct.pack("<" + fmt, v)

    def scalar(self, fmt, size, v):
        self.prep(size, 0)
        self._prepend(fmt, v)

    def uoffset_to(self, off):
        self.prep(4, 0)
        self._prepend("I", len(self.b) - off + 4)

    # ---- vectors
    def vector_bytes(self, data):
        self.prep(4, len(data))
        self.b[0:0] = bytes(data)
        self._prepend("I", len(data))
        return self.offset()

    def vector_u64(self, vals):
        self.prep(4, 8 * len(vals)); self.prep(8, 8 * len(vals))
        for v in reversed(vals):
            self._prepend("Q", v)
        self._prepend("I", len(vals))
        return self.offset()

    def vector_offsets(self, offs):
        self.prep(4, 4 * len(offs))
        for o in reversed(offs):
            self.uoffset_to(o)
        self._prepend("I", len(offs))
        return self.offset()

    # ---- tables: fields = {slot: ("offset", off) | ("u64", v) | ("u8", v)}
    def table(self, nslots, fields):
        start = self.offset()
        slot_off = [0] * nslots
        # official generated code adds fields in declaration order of decreasing size; any order is valid FlatBuffers
        for slot in sorted(fields, key=lambda s: {"u64": 0, "offset": 1, "u8": 2}[fields[s][0]]):
            kind, v = fields[slot]
            if kind == "offset":
                self.uoffset_to(v)
            elif kind == "u64":
                self.scalar("Q", 8, v)
            else:
                self.scalar("B", 1, v)
            slot_off[slot] = self.offset()
        self.prep(4, 0)
        self._prepend("i", 0)                                   # soffset placeholder
        obj = self.offset()
        vt = [4 + 2 * nslots, obj - start] + [(obj - o) if o else 0 for o in slot_off]
        key = tuple(vt)
        if key in self.vtables:                                  # vtable sharing, as the official builders do
            vt_off = self.vtables[key]
        else:
            for x in reversed(vt):
                self._prepend("H", x)
            vt_off = self.offset()
            self.vtables[key] = vt_off
        pos = len(self.b) - obj                                  # the table's soffset field, from the current front
        struct.pack_into("<i", self.b, pos, vt_off - obj)
        return obj

    def finish_size_prefixed(self, root, ident=b"zkif"):
        self.prep(self.minalign, 4 + 4 + 4)
        self.b[0:0] = ident
        self.uoffset_to(root)
        self._prepend("I", len(self.b))
        return bytes(self.b)


# ---- zkInterface messages ------------------------------------------------------------------------
def _variables(B, ids, values, width):
    vals = b"".join(int(v).to_bytes(width, "little") for v in values) if values is not None else None
    voff = B.vector_bytes(vals) if vals is not None else None
    ioff = B.vector_u64(ids)
    f = {0: ("offset", ioff)}
    if voff is not None:
        f[1] = ("offset", voff)
    return B.table(3, f)


def circuit_header(instance_ids, instance_values, free_variable_id, field_maximum, width=32):
    B = Builder()
    fm = B.vector_bytes(int(field_maximum).to_bytes(32, "little"))
    iv = _variables(B, instance_ids, instance_values, width)
    hdr = B.table(4, {0: ("offset", iv), 1: ("u64", free_variable_id), 2: ("offset", fm)})
    root = B.table(2, {0: ("u8", 1), 1: ("offset", hdr)})
    return B.finish_size_prefixed(root)


def constraint_system(constraints, width=32):
    """constraints: list of (a, b, c), each a list of (variable id, coefficient)"""
    B = Builder()
    offs = []
    for lcs in constraints:
        sub = [_variables(B, [i for i, _ in lc], [v for _, v in lc], width) for lc in reversed(lcs)][::-1]
        offs.append(B.table(3, {0: ("offset", sub[0]), 1: ("offset", sub[1]), 2: ("offset", sub[2])}))
    vec = B.vector_offsets(offs)
    cs = B.table(2, {0: ("offset", vec)})
    root = B.table(2, {0: ("u8", 2), 1: ("offset", cs)})
    return B.finish_size_prefixed(root)


def witness(ids, values, width=32):
    B = Builder()
    av = _variables(B, ids, values, width)
    w = B.table(1, {0: ("offset", av)})
    root = B.table(2, {0: ("u8", 3), 1: ("offset", w)})
    return B.finish_size_prefixed(root)
