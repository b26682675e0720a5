#!/usr/bin/env python
"""Packed descriptor wire path: the step either side of the matching hot path.

The reference keeps one `GlobalDescriptor` message per keyframe in a SortedDict, each holding the
descriptor as a Python list of floats (`embedding.tolist()`, gdlcd.py:164-168), publishes them in
chunks of `frontend.detection_publication_max_elems_per_msg` (gdlcd.py:198-227 through
utils/misc.py:21-32 `dict_to_list_chunks`) and turns every received list back into a float64 array
(`np.asarray(msg.descriptor)`, lcsm.py:63).  Here the buffer is one packed float32 array:

    PackedDescriptorBuffer   append / chunks / delete_below   <- gdlcd.py:164-168, 176-185, 198-227
    DescriptorChunk          to_bytes / from_bytes / messages <- one GlobalDescriptors message
    unknown_rows             which rows of a received chunk are new <- neighbors_manager.py:147-169

Precision contract (unchanged): float32 on the wire (the message field is float32[]), float64 after
reception -- `DescriptorChunk.as_float64()` is what `process_remote_descriptors` is fed, so the bank
rows and the query values are exactly the reference's.
"""
import struct
from collections import namedtuple
from types import SimpleNamespace

import numpy as np

_MAGIC = b"CSGD"
_HEADER = struct.Struct("<4sHHiii")        # magic, version, flags, robot_id, count, dim


class DescriptorChunk(namedtuple("DescriptorChunk", ["robot_id", "keyframe_ids", "descriptors"])):
    """One GlobalDescriptors message: `descriptors` float32 [m, dim], `keyframe_ids` int32 [m]."""
    __slots__ = ()

    def __len__(self):
        return len(self.keyframe_ids)

    def nbytes_payload(self):
        """What gdlcd.py:214-219 logs as communication: elements x dim x 4 bytes."""
        return int(self.descriptors.shape[0] * self.descriptors.shape[1] * 4)

    def as_float64(self):
        """The values the receiver works with (np.asarray of a float32[] field is float64)."""
        return self.descriptors.astype(np.float64)

    def to_bytes(self):
        d = np.ascontiguousarray(self.descriptors, dtype="<f4")
        k = np.ascontiguousarray(self.keyframe_ids, dtype="<i4")
        return _HEADER.pack(_MAGIC, 1, 0, int(self.robot_id), d.shape[0], d.shape[1]) + k.tobytes() + d.tobytes()

    @classmethod
    def from_bytes(cls, buf):
        magic, version, _, robot_id, count, dim = _HEADER.unpack_from(buf, 0)
        if magic != _MAGIC or version != 1:
            raise ValueError("not a packed descriptor chunk")
        need = _HEADER.size + count * 4 + count * dim * 4
        if count < 0 or dim < 0 or len(buf) != need:
            raise ValueError(f"packed descriptor chunk: expected {need} bytes, got {len(buf)}")
        off = _HEADER.size
        k = np.frombuffer(buf, dtype="<i4", count=count, offset=off).astype(np.int32)
        d = np.frombuffer(buf, dtype="<f4", count=count * dim, offset=off + 4 * count)
        return cls(robot_id, k, d.reshape(count, dim).astype(np.float32))

    def messages(self):
        """Duck-typed GlobalDescriptor messages (robot_id, keyframe_id, descriptor) for callers that
        still take them one by one (lcsm.py:56-72)."""
        return [SimpleNamespace(robot_id=int(self.robot_id), keyframe_id=int(k), descriptor=row.tolist())
                for k, row in zip(self.keyframe_ids, self.descriptors)]


class PackedDescriptorBuffer(object):
    """The publication buffer `global_descriptors_buffer` (gdlcd.py:103, 164-168) as packed arrays,
    ordered by keyframe id like the reference's SortedDict."""

    def __init__(self, robot_id, dim=None, capacity=1024):
        self.robot_id = int(robot_id)
        self.dim = dim
        self._ids = np.empty(capacity, dtype=np.int64)
        self._rows = None if dim is None else np.empty((capacity, dim), dtype=np.float32)
        self._n = 0

    def __len__(self):
        return self._n

    @property
    def keyframe_ids(self):
        return self._ids[:self._n]

    @property
    def descriptors(self):
        return self._rows[:self._n] if self._rows is not None else np.zeros((0, 0), np.float32)

    def first_key(self):
        return int(self._ids[0])

    def last_key(self):
        return int(self._ids[self._n - 1])

    def _reserve(self, extra, dim):
        if self._rows is None:
            self.dim = int(dim)
            self._rows = np.empty((len(self._ids), self.dim), dtype=np.float32)
        if dim != self.dim:
            raise ValueError(f"descriptor of length {dim} in a buffer of dimension {self.dim}")
        need = self._n + extra
        if need > len(self._ids):
            cap = len(self._ids)
            while cap < need:
                cap *= 2
            self._ids = np.resize(self._ids, cap)
            rows = np.empty((cap, self.dim), dtype=np.float32)
            rows[:self._n] = self._rows[:self._n]
            self._rows = rows

    def append(self, keyframe_id, embedding):
        """gdlcd.py:164-168: store one descriptor (cast to the float32 of the message field);
        an existing keyframe id is overwritten, like the dict assignment."""
        e = np.asarray(embedding)
        assert e.ndim == 1
        self._reserve(1, e.shape[0])
        k = int(keyframe_id)
        pos = int(np.searchsorted(self._ids[:self._n], k))
        if pos < self._n and self._ids[pos] == k:
            self._rows[pos] = e
            return
        if pos < self._n:
            self._ids[pos + 1:self._n + 1] = self._ids[pos:self._n].copy()
            self._rows[pos + 1:self._n + 1] = self._rows[pos:self._n].copy()
        self._ids[pos] = k
        self._rows[pos] = e
        self._n += 1

    def extend(self, keyframe_ids, embeddings):
        """append() for a batch; new keyframe ids arriving in ascending order after the last stored one (the
        normal case: keyframes are produced in order) are copied in one block."""
        e = np.asarray(embeddings)
        k = np.asarray(keyframe_ids, dtype=np.int64)
        if e.ndim == 2 and len(k) == e.shape[0] and len(k) > 0 and np.all(k[1:] > k[:-1]) and \
                (self._n == 0 or k[0] > self._ids[self._n - 1]):
            self._reserve(len(k), e.shape[1])
            self._ids[self._n:self._n + len(k)] = k
            self._rows[self._n:self._n + len(k)] = e
            self._n += len(k)
            return
        for ki, ei in zip(k.tolist(), e):
            self.append(ki, ei)

    def chunks(self, start, chunk_size):
        """`dict_to_list_chunks(buffer, start, chunk_size)` (utils/misc.py:21-32): the entries whose
        key is >= start, in key order, cut into chunks of at most chunk_size."""
        lo = int(np.searchsorted(self._ids[:self._n], int(start), side="left"))
        out = []
        for a in range(lo, self._n, int(chunk_size)):
            b = min(a + int(chunk_size), self._n)
            out.append(DescriptorChunk(self.robot_id, self._ids[a:b].astype(np.int32), self._rows[a:b].copy()))
        return out

    def delete_below(self, from_kf_id):
        """gdlcd.py:176-185 `delete_useless_descriptors`: drop keys < from_kf_id, but only when
        from_kf_id >= the first key."""
        if self._n == 0 or from_kf_id < self._ids[0]:
            return 0
        lo = int(np.searchsorted(self._ids[:self._n], int(from_kf_id), side="left"))
        if lo:
            self._ids[:self._n - lo] = self._ids[lo:self._n].copy()
            self._rows[:self._n - lo] = self._rows[lo:self._n].copy()
            self._n -= lo
        return lo


def unknown_rows(chunk, last_keyframe_received):
    """neighbors_manager.py:147-169 `get_unknown_range`: indexes of the rows of a received chunk
    whose keyframe id is newer than the last one received from that robot, and the updated
    last-received id."""
    ids = np.asarray(chunk.keyframe_ids)
    rows = np.nonzero(ids > last_keyframe_received)[0]
    last = max(int(last_keyframe_received), int(ids.max())) if len(ids) else int(last_keyframe_received)
    return rows, last
