"""h5min - a minimal HDF5 writer (and a reader for what it writes): the on-disk side of the reference's outputs without an HDF5
library in the image (src/smc_main.jl:513-526: `particle_store_path` holds the HDF5 dataset `smcparams`, `savepath` the objects
`cloud`, `w`, `W`).

Scope: float64 / int64 arrays of any rank and scalars, in the root group and in nested groups; contiguous layout, no chunking,
no compression, no attributes.  File layout (HDF5 File Format Specification v1.x objects, the ones every libhdf5 reads):
superblock version 0 -> root group (object header v1 with a Symbol Table message) -> per group one v1 B-tree node ("TREE") with one
symbol-table node ("SNOD", sorted by name) and a local heap ("HEAP") for the names -> per dataset an object header with Dataspace
(v1), Datatype (v1) and Data Layout (v3, contiguous) messages.

Julia convention (HDF5.jl / JLD2): a column-major Julia array of size (n1, n2, ...) is stored as a dataset of dims (..., n2, n1)
holding the same bytes.  `write_julia` applies it: a numpy array of shape (N, d) comes back in Julia as an N x d Matrix{Float64} -
what `h5open(particle_store_path)["smcparams"]` and JLD2's `load(savepath, "w")` return in the reference."""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIG = b"\x89HDF\r\n\x1a\n"
LEAF_K, INTERNAL_K = 32, 16          # group leaf node K: a symbol-table node holds up to 2 K = 64 entries


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _msg(mtype, data):
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), 0) + data


def _dtype_msg(dt):
    if dt == np.float64:      # class 1 (floating point), version 1; little endian, msb of the mantissa implied, sign bit 63
        return _msg(0x0003, struct.pack("<B3BI", 0x11, 0x20, 0x3F, 0x00, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023))
    if dt == np.int64:        # class 0 (fixed point), version 1; little endian, two's complement
        return _msg(0x0003, struct.pack("<B3BI", 0x10, 0x08, 0x00, 0x00, 8) + struct.pack("<HH", 0, 64))
    raise TypeError("h5min writes float64 and int64 only")


class _Writer:
    def __init__(self):
        self.buf = bytearray()

    def alloc(self, data, align=8):
        self.buf += b"\0" * (-len(self.buf) % align)
        addr = len(self.buf)
        self.buf += data
        return addr

    def dataset(self, arr):
        arr = np.asarray(arr)
        arr = arr.copy(order="C") if arr.ndim else arr                # (ascontiguousarray would turn a scalar into a 1-vector)
        if arr.dtype not in (np.float64, np.int64):
            arr = arr.astype(np.int64 if np.issubdtype(arr.dtype, np.integer) or arr.dtype == bool else np.float64)
        raw = arr.tobytes()
        daddr = self.alloc(raw) if raw else UNDEF
        space = struct.pack("<BBB5x", 1, arr.ndim, 0) + b"".join(struct.pack("<Q", s) for s in arr.shape)
        msgs = _msg(0x0001, space) + _dtype_msg(arr.dtype) + _msg(0x0008, struct.pack("<BBQQ", 3, 1, daddr, len(raw)))
        return self.alloc(struct.pack("<BxHII4x", 1, 3, 1, len(msgs)) + msgs)

    def group(self, items):
        """items: dict name -> ndarray | dict (subgroup).  Returns (object header address, btree address, heap address)."""
        names = sorted(items, key=lambda s: s.encode())
        if len(names) > 2 * LEAF_K:
            raise ValueError("h5min: more than %d entries in one group" % (2 * LEAF_K))
        entries = []
        for nm in names:
            v = items[nm]
            entries.append(self.group(v)[0] if isinstance(v, dict) else self.dataset(v))
        # local heap: "" at offset 0, then the names, then one free block
        heap = bytearray(b"\0" * 8)
        offs = []
        for nm in names:
            offs.append(len(heap))
            heap += _pad8(nm.encode() + b"\0")
        free_off = len(heap)
        heap += struct.pack("<QQ", 1, 32) + b"\0" * 16            # free block: next = 1 (last), size 32
        heap_data = self.alloc(bytes(heap))
        heap_addr = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), free_off, heap_data))
        snod = b"SNOD" + struct.pack("<BxH", 1, len(names))
        for off, addr in zip(offs, entries):
            snod += struct.pack("<QQII16x", off, addr, 0, 0)
        snod += b"\0" * (8 + 2 * LEAF_K * 40 - len(snod))
        snod_addr = self.alloc(snod)
        tree = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if names else 0, UNDEF, UNDEF)
        tree += struct.pack("<QQQ", 0, snod_addr, offs[-1] if offs else 0)
        tree += b"\0" * (24 + (2 * INTERNAL_K + 1) * 8 + 2 * INTERNAL_K * 8 - len(tree))
        tree_addr = self.alloc(tree)
        msgs = _msg(0x0011, struct.pack("<QQ", tree_addr, heap_addr))
        hdr = self.alloc(struct.pack("<BxHII4x", 1, 1, 1, len(msgs)) + msgs)
        return hdr, tree_addr, heap_addr


def write(path, items):
    """Write `items` (dict name -> array / scalar / nested dict) as an HDF5 file; arrays keep numpy's (C-order) dims."""
    w = _Writer()
    w.buf += b"\0" * 96                                            # superblock goes here
    hdr, tree, heap = w.group(items)
    eof = len(w.buf)
    sb = SIG + struct.pack("<8B", 0, 0, 0, 0, 0, 8, 8, 0) + struct.pack("<HHI", LEAF_K, INTERNAL_K, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    sb += struct.pack("<QQII", 0, hdr, 1, 0) + struct.pack("<QQ", tree, heap)      # root symbol-table entry (cached: B-tree, heap)
    assert len(sb) == 96
    w.buf[:96] = sb
    with open(path, "wb") as f:                                    # the exact path, whatever its extension
        f.write(bytes(w.buf))


def _julia(v):
    if isinstance(v, dict):
        return {k: _julia(x) for k, x in v.items()}
    a = np.asarray(v)
    return np.ascontiguousarray(a.T) if a.ndim > 1 else a          # dims reversed, bytes = the column-major Julia array


def write_julia(path, items):
    """As `write`, with the HDF5.jl / JLD2 array convention: a numpy array of shape (n1, n2) is read back in Julia as n1 x n2."""
    write(path, _julia(items))


# ------------------------------------------------------------------------------------------------ reader (for files written above)
def read(path, julia=False):
    """Read a file written by this module back into a dict (nested dicts for groups).  julia = True undoes write_julia."""
    with open(path, "rb") as f:
        b = f.read()
    if b[:8] != SIG or b[8] != 0:
        raise ValueError("not an HDF5 file with a version-0 superblock")
    root_hdr = struct.unpack_from("<Q", b, 24 + 32 + 8)[0]

    def messages(addr):
        ver, nmsg, _, size = struct.unpack_from("<BxHII", b, addr)
        assert ver == 1
        p, out = addr + 16, []
        for _ in range(nmsg):
            mtype, msize = struct.unpack_from("<HH", b, p)
            out.append((mtype, b[p + 8:p + 8 + msize]))
            p += 8 + msize
        return out

    def obj(addr):
        msgs = dict(messages(addr))
        if 0x0011 in msgs:
            tree, heap = struct.unpack("<QQ", msgs[0x0011][:16])
            hsize, _, hdata = struct.unpack_from("<QQQ", b, heap + 8)
            used = struct.unpack_from("<H", b, tree + 6)[0]
            out = {}
            for e in range(used):
                snod = struct.unpack_from("<Q", b, tree + 24 + 8 + 16 * e)[0]
                assert b[snod:snod + 4] == b"SNOD"
                for k in range(struct.unpack_from("<H", b, snod + 6)[0]):
                    noff, oaddr = struct.unpack_from("<QQ", b, snod + 8 + 40 * k)
                    name = b[hdata + noff:b.index(b"\0", hdata + noff)].decode()
                    out[name] = obj(oaddr)
            return out
        sp, dt, lay = msgs[0x0001], msgs[0x0003], msgs[0x0008]
        rank = sp[1]
        dims = struct.unpack_from("<%dQ" % rank, sp, 8) if rank else ()
        dtype = np.float64 if (dt[0] & 0x0F) == 1 else np.int64
        _, cls, daddr, dsize = struct.unpack_from("<BBQQ", lay, 0)
        assert cls == 1
        a = np.frombuffer(b, dtype=dtype, count=dsize // 8, offset=daddr).reshape(dims) if dsize else np.zeros(dims, dtype)
        if julia and a.ndim > 1:
            a = a.T
        return a.copy()

    return obj(root_hdr)
