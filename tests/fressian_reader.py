"""A small Fressian READER, written for the tests from the published format description of org.fressian (the library behind
clojure.data.fressian, which maelstrom.net.journal uses, journal.clj:24,55-114) — independent of the tables in
maelstrom_amd/csrc/fressian.cpp.  TEST INFRASTRUCTURE.  Supports what a net journal contains: ints, strings, keywords, nil,
booleans, lists (packed, counted, closed / open), maps, user structs ("ev", "msg") and both caches."""
import io
import struct


class Keyword(str):
    def __repr__(self):
        return ":" + str(self)


class Struct:
    def __init__(self, tag, fields):
        self.tag, self.fields = tag, fields

    def __repr__(self):
        return f"#{self.tag}{self.fields}"


_END = object()
_UNDER_CONSTRUCTION = object()


class Reader:
    def __init__(self, data):
        self.f = io.BytesIO(data)
        self.priority, self.structs = [], []

    def _byte(self):
        b = self.f.read(1)
        if not b:
            raise EOFError
        return b[0]

    def _raw(self, n):
        b = self.f.read(n)
        if len(b) != n:
            raise EOFError
        return b

    def _packed(self, code, zero, nbytes):
        # the lead byte carries the (signed) high bits relative to the ZERO code, the low `nbytes` bytes follow big-endian
        return ((code - zero) << (8 * nbytes)) | int.from_bytes(self._raw(nbytes), "big")

    def read_int(self):
        v = self.read()
        assert isinstance(v, int) and not isinstance(v, bool), v
        return v

    def read(self):
        c = self._byte()
        if c <= 0x3F:
            return c
        if c == 0xFF:
            return -1
        if 0x40 <= c < 0x60:
            return self._packed(c, 0x50, 1)
        if 0x60 <= c < 0x70:
            return self._packed(c, 0x68, 2)
        if 0x70 <= c < 0x74:
            return self._packed(c, 0x72, 3)
        if 0x74 <= c < 0x78:
            return self._packed(c, 0x76, 4)
        if 0x78 <= c < 0x7C:
            return self._packed(c, 0x7A, 5)
        if 0x7C <= c < 0x80:
            return self._packed(c, 0x7E, 6)
        if c == 0xF8:
            return struct.unpack(">q", self._raw(8))[0]
        if 0x80 <= c < 0xA0:
            return self._cached(c - 0x80)
        if c == 0xCC:
            return self._cached(self.read_int())
        if c == 0xCD:   # put priority cache: the slot is taken before the object's components are read
            idx = len(self.priority)
            self.priority.append(_UNDER_CONSTRUCTION)
            o = self.read()
            self.priority[idx] = o
            return o
        if 0xA0 <= c < 0xB0:
            return self._struct(*self.structs[c - 0xA0])
        if c == 0xF0:
            return self._struct(*self.structs[self.read_int()])
        if c == 0xEF:
            tag, n = self.read(), self.read_int()
            self.structs.append((tag, n))
            return self._struct(tag, n)
        if 0xDA <= c < 0xE2:
            return self._raw(c - 0xDA).decode("utf-8")
        if c == 0xE3:
            return self._raw(self.read_int()).decode("utf-8")
        if 0xE4 <= c < 0xEC:
            return [self.read() for _ in range(c - 0xE4)]
        if c == 0xEC:
            return [self.read() for _ in range(self.read_int())]
        if c in (0xED, 0xEE):   # closed / open list: until END_COLLECTION (an open list may also end at EOF)
            out = []
            while True:
                try:
                    o = self.read()
                except EOFError:
                    if c == 0xEE:
                        return out
                    raise
                if o is _END:
                    return out
                out.append(o)
        if c == 0xFD:
            return _END
        if c == 0xC0:
            kvs = self.read()
            assert len(kvs) % 2 == 0
            return dict(zip(kvs[0::2], kvs[1::2]))
        if c == 0xCA:
            ns, name = self.read(), self.read()
            return Keyword(name if ns is None else f"{ns}/{name}")
        if c == 0xF7:
            return None
        if c == 0xF5:
            return True
        if c == 0xF6:
            return False
        raise ValueError(f"unsupported Fressian code {c:#x} at offset {self.f.tell() - 1}")

    def _cached(self, idx):
        o = self.priority[idx]
        assert o is not _UNDER_CONSTRUCTION
        return o

    def _struct(self, tag, n):
        return Struct(tag, [self.read() for _ in range(n)])


def read_journal(data):
    """bytes of a net-journal stripe -> list of {:id :time :type :message {:id :src :dest :body}} dicts (journal.clj:53, message.clj:8)"""
    r = Reader(data)
    out = []
    while True:
        try:
            ev = r.read()
        except EOFError:
            return out
        assert isinstance(ev, Struct) and ev.tag == "ev" and len(ev.fields) == 4, ev
        msg = ev.fields[3]
        assert isinstance(msg, Struct) and msg.tag == "msg" and len(msg.fields) == 4, msg
        out.append({"id": ev.fields[0], "time": ev.fields[1], "type": ev.fields[2],
                    "message": {"id": msg.fields[0], "src": msg.fields[1], "dest": msg.fields[2], "body": msg.fields[3]}})
