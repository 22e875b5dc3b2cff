"""Minimal read-only HDF5 parser for the subset Keras-3 writes into ``model.weights.h5``.

Keras 3 saves weights through h5py with the library's oldest on-disk format:
superblock version 0, version-1 object headers, groups stored as symbol
tables (v1 B-tree + local heap + SNOD nodes) and every dataset contiguous,
little-endian IEEE float32, unfiltered.  That subset is all this module reads;
anything else raises ``Hdf5FormatError`` so a silently wrong tensor can never
reach the sampler.  No third-party HDF5 library is needed (none is installed
on the GPU box).

Only ``read_datasets(buf)`` is public: it returns ``{"/group/.../name": ndarray}``.
"""
from __future__ import annotations

import struct
from typing import Dict, Tuple

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF

_MSG_DATASPACE = 0x0001
_MSG_DATATYPE = 0x0003
_MSG_LAYOUT = 0x0008
_MSG_CONTINUATION = 0x0010
_MSG_SYMBOL_TABLE = 0x0011


class Hdf5FormatError(ValueError):
    """The file uses an HDF5 feature outside the supported subset."""


class _Reader:
    def __init__(self, buf: bytes):
        self.buf = memoryview(buf)
        if bytes(self.buf[:8]) != _SIG:
            raise Hdf5FormatError("not an HDF5 file (bad signature)")
        ver = self.buf[8]
        if ver not in (0, 1):
            raise Hdf5FormatError(f"superblock version {ver} not supported (need 0 or 1)")
        self.so = self.buf[13]  # size of offsets
        self.sl = self.buf[14]  # size of lengths
        if self.so != 8 or self.sl != 8:
            raise Hdf5FormatError("only 8-byte offsets/lengths supported")
        pos = 24 + (4 if ver == 1 else 0)
        self.base, _free, _eof, _drv = struct.unpack_from("<4Q", self.buf, pos)
        pos += 32
        # root group symbol table entry
        self.root_header = struct.unpack_from("<Q", self.buf, pos + 8)[0]

    # -- primitives -----------------------------------------------------
    def u(self, pos: int, fmt: str):
        return struct.unpack_from("<" + fmt, self.buf, self.base + pos)

    def messages(self, addr: int):
        """Yield (type, body_offset, body_size) for a version-1 object header."""
        version, _r, nmsgs, _ref, hsize = self.u(addr, "BBHII")
        if version != 1:
            raise Hdf5FormatError(f"object header version {version} not supported")
        blocks = [(addr + 16, hsize)]
        seen = 0
        while blocks and seen < nmsgs:
            pos, size = blocks.pop(0)
            end = pos + size
            while pos + 8 <= end and seen < nmsgs:
                mtype, msize, _flags = self.u(pos, "HHB")
                body = pos + 8
                seen += 1
                if mtype == _MSG_CONTINUATION:
                    coff, clen = self.u(body, "QQ")
                    blocks.append((coff, clen))
                else:
                    yield mtype, body, msize
                pos = body + msize  # v1 message sizes are already multiples of 8

    def heap_string(self, heap_addr: int, offset: int) -> str:
        if bytes(self.buf[self.base + heap_addr:self.base + heap_addr + 4]) != b"HEAP":
            raise Hdf5FormatError("bad local heap signature")
        data_addr = self.u(heap_addr + 24, "Q")[0]
        start = self.base + data_addr + offset
        end = start
        while self.buf[end] != 0:
            end += 1
        return bytes(self.buf[start:end]).decode("utf-8")

    def group_entries(self, btree_addr: int, heap_addr: int):
        """Yield (name, object_header_addr) for every link below a v1 group B-tree."""
        sig = bytes(self.buf[self.base + btree_addr:self.base + btree_addr + 4])
        if sig != b"TREE":
            raise Hdf5FormatError("bad B-tree signature")
        ntype, level, used = self.u(btree_addr + 4, "BBH")
        if ntype != 0:
            raise Hdf5FormatError("expected a group B-tree node")
        pos = btree_addr + 8 + 16  # skip siblings
        pos += 8  # key 0
        for _ in range(used):
            child = self.u(pos, "Q")[0]
            pos += 16  # child + next key
            if level > 0:
                yield from self.group_entries(child, heap_addr)
            else:
                yield from self.snod_entries(child, heap_addr)

    def snod_entries(self, addr: int, heap_addr: int):
        if bytes(self.buf[self.base + addr:self.base + addr + 4]) != b"SNOD":
            raise Hdf5FormatError("bad symbol node signature")
        nsym = self.u(addr + 6, "H")[0]
        pos = addr + 8
        for _ in range(nsym):
            name_off, hdr = self.u(pos, "QQ")
            yield self.heap_string(heap_addr, name_off), hdr
            pos += 40

    # -- objects --------------------------------------------------------
    def walk(self, header_addr: int, prefix: str, out: Dict[str, np.ndarray]):
        shape = None
        dtype = None
        layout = None
        symtab = None
        for mtype, body, _size in self.messages(header_addr):
            if mtype == _MSG_SYMBOL_TABLE:
                symtab = self.u(body, "QQ")
            elif mtype == _MSG_DATASPACE:
                shape = self._dataspace(body)
            elif mtype == _MSG_DATATYPE:
                dtype = self._datatype(body)
            elif mtype == _MSG_LAYOUT:
                layout = self._layout(body)
        if symtab is not None:
            for name, hdr in self.group_entries(*symtab):
                self.walk(hdr, prefix + "/" + name, out)
            return
        if shape is None or dtype is None or layout is None:
            if shape is None and dtype is None and layout is None:
                return  # empty group stored without a symbol table
            raise Hdf5FormatError(f"{prefix}: incomplete dataset header")
        kind, a, b = layout
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        nbytes = count * dtype.itemsize
        if kind == "contiguous":
            if a == _UNDEF:
                data = np.zeros(shape, dtype)
            else:
                if b < nbytes:
                    raise Hdf5FormatError(f"{prefix}: layout smaller than dataspace")
                data = np.frombuffer(self.buf, dtype, count, self.base + a).reshape(shape)
        else:  # compact
            data = np.frombuffer(self.buf, dtype, count, self.base + a).reshape(shape)
        out[prefix] = np.array(data)  # own the memory

    def _dataspace(self, body: int) -> Tuple[int, ...]:
        version, ndim, flags = self.u(body, "BBB")
        if version == 1:
            pos = body + 8
        elif version == 2:
            pos = body + 4
        else:
            raise Hdf5FormatError(f"dataspace version {version}")
        return tuple(self.u(pos, f"{ndim}Q")) if ndim else ()

    def _datatype(self, body: int) -> np.dtype:
        cv, b0, _b1, _b2, size = self.u(body, "BBBBI")
        cls = cv & 0x0F
        if cls == 1 and size in (4, 8) and (b0 & 1) == 0:
            return np.dtype("<f4" if size == 4 else "<f8")
        if cls == 0 and size in (1, 2, 4, 8) and (b0 & 1) == 0:
            signed = (b0 >> 3) & 1
            return np.dtype(("<i" if signed else "<u") + str(size))
        raise Hdf5FormatError(f"datatype class {cls} size {size} not supported")

    def _layout(self, body: int):
        version, cls = self.u(body, "BB")
        if version != 3:
            raise Hdf5FormatError(f"layout version {version} not supported")
        if cls == 1:
            addr, size = self.u(body + 2, "QQ")
            return "contiguous", addr, size
        if cls == 0:
            size = self.u(body + 2, "H")[0]
            return "compact", body + 4, size
        raise Hdf5FormatError("chunked datasets not supported")


def read_datasets(buf: bytes) -> Dict[str, np.ndarray]:
    """Return every dataset of an HDF5 image as ``{absolute_path: ndarray}``."""
    r = _Reader(buf)
    out: Dict[str, np.ndarray] = {}
    r.walk(r.root_header, "", out)
    return out
