"""TensorBoard event files without TensorFlow: the writer behind TrainerController's epoch scalars
(src/Network/TrainerController.py:181-182 `tf.summary.create_file_writer`, :396-412 `_update_summary_logging`).

File format (TFRecord framing + two tiny protobuf messages, all written by hand here):
    record  = uint64 length | uint32 masked_crc32c(length bytes) | data | uint32 masked_crc32c(data)
    data    = Event { 1: double wall_time, 2: int64 step, 3: string file_version | 5: Summary summary }
    Summary = { 1: repeated Value { 1: string tag, 8: TensorProto tensor, 9: SummaryMetadata metadata } }
`tf.summary.scalar` of TF 2.x stores the value as a rank-0 DT_FLOAT TensorProto (dtype, empty shape, 4 bytes of
tensor_content) with metadata.plugin_data.plugin_name = "scalars"; that is what `scalar()` emits, so TensorBoard shows the
files exactly like the reference's.  The first record of a file is Event{wall_time, file_version: "brain.Event:2"}.
`read_events` parses such files back (crc-checked); it also understands the older `simple_value` form."""
import os
import socket
import struct
import time

_CRC_TABLE = []


def _crc_table():
    if not _CRC_TABLE:
        poly = 0x82F63B78                      # CRC-32C (Castagnoli), reflected
        for n in range(256):
            c = n
            for _ in range(8):
                c = (c >> 1) ^ poly if c & 1 else c >> 1
            _CRC_TABLE.append(c)
    return _CRC_TABLE


def crc32c(data):
    t = _crc_table()
    c = 0xFFFFFFFF
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---------------------------------------------------------------- protobuf wire encoding (the few types needed)
def _varint(n):
    n &= (1 << 64) - 1                          # int64 two's complement, as protobuf encodes negative values
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field, wire):
    return _varint((field << 3) | wire)


def _ld(field, payload):                        # length-delimited field
    return _key(field, 2) + _varint(len(payload)) + payload


def _scalar_value(tag, value):
    tensor = _key(1, 0) + _varint(1) + _ld(2, b"") + _ld(4, struct.pack("<f", float(value)))   # DT_FLOAT, shape {}, content
    metadata = _ld(1, _ld(1, b"scalars"))                                                      # plugin_data.plugin_name
    return _ld(1, tag.encode("utf-8")) + _ld(8, tensor) + _ld(9, metadata)


def _event(wall_time, step=None, file_version=None, summary=None):
    e = _key(1, 1) + struct.pack("<d", wall_time)
    if step is not None:
        e += _key(2, 0) + _varint(int(step))
    if file_version is not None:
        e += _ld(3, file_version.encode("ascii"))
    if summary is not None:
        e += _ld(5, summary)
    return e


class SummaryWriter:
    """tf.summary.create_file_writer(logdir) + tf.summary.scalar(tag, value, step) for float scalars."""

    _seq = 0

    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        SummaryWriter._seq += 1
        now = time.time()
        name = "events.out.tfevents.%010d.%s.%d.%d.v2" % (int(now), socket.gethostname(), os.getpid(), SummaryWriter._seq)
        self.path = os.path.join(logdir, name)
        self._f = open(self.path, "wb")
        self._record(_event(now, file_version="brain.Event:2"))
        self.flush()

    def _record(self, data):
        head = struct.pack("<Q", len(data))
        self._f.write(head + struct.pack("<I", masked_crc32c(head)) + data + struct.pack("<I", masked_crc32c(data)))

    def scalar(self, tag, value, step):
        self._record(_event(time.time(), step=step, summary=_ld(1, _scalar_value(tag, value))))

    def flush(self):
        self._f.flush()

    def close(self):
        if not self._f.closed:
            self._f.close()


# ---------------------------------------------------------------- reader (tests, tools)
def _read_varint(buf, pos):
    shift = n = 0
    while True:
        b = buf[pos]
        pos += 1
        n |= (b & 0x7F) << shift
        if not b & 0x80:
            return n, pos
        shift += 7


def _fields(buf):
    """Yield (field, wire, value) of one serialized message; value is an int (varint), bytes (fixed / length-delimited)."""
    pos = 0
    while pos < len(buf):
        k, pos = _read_varint(buf, pos)
        field, wire = k >> 3, k & 7
        if wire == 0:
            v, pos = _read_varint(buf, pos)
        elif wire == 1:
            v, pos = buf[pos:pos + 8], pos + 8
        elif wire == 5:
            v, pos = buf[pos:pos + 4], pos + 4
        elif wire == 2:
            n, pos = _read_varint(buf, pos)
            v, pos = buf[pos:pos + n], pos + n
        else:
            raise ValueError("unsupported wire type %d" % wire)
        yield field, wire, v


def read_events(path):
    """-> list of dicts {wall_time, step, file_version?, scalars: {tag: float}, plugins: {tag: plugin_name}}.
    Raises ValueError on a framing or crc error."""
    with open(path, "rb") as f:
        buf = f.read()
    out = []
    pos = 0
    while pos < len(buf):
        head = buf[pos:pos + 8]
        if len(head) < 8 or pos + 12 > len(buf):
            raise ValueError("truncated record header at %d" % pos)
        (n,) = struct.unpack("<Q", head)
        (c,) = struct.unpack("<I", buf[pos + 8:pos + 12])
        if c != masked_crc32c(head):
            raise ValueError("length crc mismatch at %d" % pos)
        data = buf[pos + 12:pos + 12 + n]
        if len(data) < n or pos + 16 + n > len(buf):
            raise ValueError("truncated record at %d" % pos)
        (c,) = struct.unpack("<I", buf[pos + 12 + n:pos + 16 + n])
        if c != masked_crc32c(data):
            raise ValueError("data crc mismatch at %d" % pos)
        pos += 16 + n
        ev = {"step": 0, "scalars": {}, "plugins": {}}
        for field, _, v in _fields(data):
            if field == 1:
                (ev["wall_time"],) = struct.unpack("<d", v)
            elif field == 2:
                ev["step"] = v - (1 << 64) if v >> 63 else v
            elif field == 3:
                ev["file_version"] = v.decode("ascii")
            elif field == 5:
                for sf, _, val in _fields(v):
                    if sf != 1:
                        continue
                    tag = value = plugin = None
                    for vf, _, vv in _fields(val):
                        if vf == 1:
                            tag = vv.decode("utf-8")
                        elif vf == 2:
                            (value,) = struct.unpack("<f", vv)
                        elif vf == 8:
                            t = dict((tf_, tv) for tf_, _, tv in _fields(vv))
                            if t.get(1) != 1:
                                raise ValueError("tensor summary %r is not DT_FLOAT" % tag)
                            if 4 in t:
                                (value,) = struct.unpack("<f", t[4])
                            elif 5 in t:
                                (value,) = struct.unpack("<f", t[5][:4])
                        elif vf == 9:
                            for mf, _, mv in _fields(vv):
                                if mf == 1:
                                    plugin = dict((pf, pv) for pf, _, pv in _fields(mv)).get(1, b"").decode("ascii")
                    ev["scalars"][tag] = value
                    if plugin is not None:
                        ev["plugins"][tag] = plugin
        out.append(ev)
    return out
