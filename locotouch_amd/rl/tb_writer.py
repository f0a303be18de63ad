"""TensorBoard event-file writer for the scalar tags the reference's runner logs
(loco_rl/loco_rl/runners/on_policy_runner.py:268-309: `Loss/*`, `Policy/mean_noise_std`, `Perf/*`, `Train/*`, `Episode*/...`).

`torch.utils.tensorboard.SummaryWriter` needs the `tensorboard` package, which this image does not have; the on-disk format
is small enough to emit directly: a TFRecord stream (length, masked CRC-32C of the length, payload, masked CRC-32C of the
payload) of `Event` protobufs - the first one carries `file_version = "brain.Event:2"`, the others a `Summary` with one
`simple_value` each.  `SummaryWriter(log_dir)` returns torch's writer when it is importable, else this one; files written
here load in TensorBoard like any other `events.out.tfevents.*` file.  `read_events` is the matching reader (tests).
"""
from __future__ import annotations

import os
import socket
import struct
import time

_CRC_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _CRC_TABLE.append(_c)


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n: int) -> bytes:
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _field(num: int, wire: int, payload: bytes) -> bytes:
    return _varint((num << 3) | wire) + payload


def _bytes_field(num: int, data: bytes) -> bytes:
    return _field(num, 2, _varint(len(data)) + data)


def _event(wall_time: float, step: int, *, file_version: str | None = None, tag: str | None = None, value: float = 0.0) -> bytes:
    ev = _field(1, 1, struct.pack("<d", wall_time)) + _field(2, 0, _varint(step))
    if file_version is not None:
        ev += _bytes_field(3, file_version.encode())
    else:
        val = _bytes_field(1, tag.encode()) + _field(2, 5, struct.pack("<f", float(value)))  # Summary.Value{tag, simple_value}
        ev += _bytes_field(5, _bytes_field(1, val))                                          # Event.summary{value}
    return ev


class EventFileWriter:
    def __init__(self, log_dir: str, flush_secs: int = 10):
        os.makedirs(log_dir, exist_ok=True)
        self.log_dir = log_dir
        name = f"events.out.tfevents.{int(time.time()):010d}.{socket.gethostname()}.{os.getpid()}.0"
        self._f = open(os.path.join(log_dir, name), "ab")
        self._flush_secs, self._last_flush = flush_secs, time.time()
        self._record(_event(time.time(), 0, file_version="brain.Event:2"))
        self.flush()

    def _record(self, data: bytes) -> None:
        head = struct.pack("<Q", len(data))
        self._f.write(head + struct.pack("<I", _masked(head)) + data + struct.pack("<I", _masked(data)))

    def add_scalar(self, tag: str, scalar_value, global_step=None, walltime=None) -> None:
        step = int(global_step) if global_step is not None else 0
        self._record(_event(time.time() if walltime is None else float(walltime), step, tag=str(tag), value=float(scalar_value)))
        if time.time() - self._last_flush > self._flush_secs:
            self.flush()

    def flush(self) -> None:
        self._f.flush()
        self._last_flush = time.time()

    def close(self) -> None:
        self.flush()
        self._f.close()


def SummaryWriter(log_dir: str, flush_secs: int = 10):
    try:
        from torch.utils.tensorboard import SummaryWriter as TorchWriter

        return TorchWriter(log_dir=log_dir, flush_secs=flush_secs)
    except Exception:  # no `tensorboard` package: write the same file format ourselves
        return EventFileWriter(log_dir, flush_secs)


def read_events(path: str) -> list[tuple[int, str, float]]:
    """(step, tag, value) of every scalar in an event file; verifies both CRCs of every record."""
    def varint(buf, i):
        n = shift = 0
        while True:
            b = buf[i]; i += 1
            n |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                return n, i

    def fields(buf):
        i = 0
        while i < len(buf):
            key, i = varint(buf, i)
            num, wire = key >> 3, key & 7
            if wire == 0:
                v, i = varint(buf, i)
            elif wire == 1:
                v, i = buf[i:i + 8], i + 8
            elif wire == 5:
                v, i = buf[i:i + 4], i + 4
            else:
                ln, i = varint(buf, i)
                v, i = buf[i:i + ln], i + ln
            yield num, wire, v

    out = []
    raw = open(path, "rb").read()
    i = 0
    while i < len(raw):
        head = raw[i:i + 8]
        (ln,) = struct.unpack("<Q", head)
        assert struct.unpack("<I", raw[i + 8:i + 12])[0] == _masked(head), "length CRC"
        data = raw[i + 12:i + 12 + ln]
        assert struct.unpack("<I", raw[i + 12 + ln:i + 16 + ln])[0] == _masked(data), "payload CRC"
        i += 16 + ln
        step, summary = 0, None
        for num, wire, v in fields(data):
            if num == 2:
                step = v
            elif num == 5:
                summary = v
        if summary is None:
            continue
        for num, _, val in fields(summary):
            if num != 1:
                continue
            tag, value = None, None
            for n2, _, v2 in fields(val):
                if n2 == 1:
                    tag = v2.decode()
                elif n2 == 2:
                    (value,) = struct.unpack("<f", v2)
            out.append((step, tag, value))
    return out
