"""Minimal read-only HDF5 reader for the reference's ``<molecule>.hdf5`` files
(OpenFermion ``MolecularData.save()``: one flat root group of small contiguous datasets).

h5py is not a dependency of this package; the molecule metadata the run path needs
(``n_electrons``, ``multiplicity``, ``n_orbitals``, ``n_qubits``, ``hf/ccsd/fci_energy`` ...,
reference src/utils/system.py:14-62) are scalars, so a ~200-line parser of the classic on-disk
format is enough: superblock v0/v1, v1 B-tree + symbol-table groups with a local heap, v1 object
headers (with continuation blocks), dataspace / datatype / layout messages, contiguous or compact
storage, fixed-point, IEEE float and fixed-length string types.  Anything else raises
``NotImplementedError`` naming the feature.
"""
import struct

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class _File:
    def __init__(self, data):
        self.d = data
        if data[:8] != _SIG:
            raise ValueError("not an HDF5 file")
        ver = data[8]
        if ver not in (0, 1):
            raise NotImplementedError(f"HDF5 superblock version {ver}")
        self.so, self.sl = data[13], data[14]          # sizes of offsets / lengths
        if self.so != 8 or self.sl != 8:
            raise NotImplementedError("HDF5 files with non-8-byte offsets")
        p = 24 if ver == 0 else 28
        self.base = struct.unpack_from("<Q", data, p)[0]
        root_entry = p + 32                             # base, free-space, eof, driver-info addresses
        self.root_header = struct.unpack_from("<Q", data, root_entry + 8)[0]
        cache_type = struct.unpack_from("<I", data, root_entry + 16)[0]
        self.root_btree = self.root_heap = None
        if cache_type == 1:
            self.root_btree, self.root_heap = struct.unpack_from("<QQ", data, root_entry + 24)

    # ---- object headers -------------------------------------------------------------------
    def messages(self, addr):
        d = self.d
        ver, _, nmsg, _, hsize = struct.unpack_from("<BBHII", d, addr)
        if ver != 1:
            raise NotImplementedError(f"object header version {ver}")
        blocks = [(addr + 16, hsize)]
        out = []
        while blocks and len(out) < nmsg:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize, _flags = struct.unpack_from("<HHB", d, p)
                body = p + 8
                if mtype == 0x10:                       # continuation
                    off, ln = struct.unpack_from("<QQ", d, body)
                    blocks.append((off, ln))
                out.append((mtype, body, msize))
                p = body + msize
        return out

    # ---- groups ------------------------------------------------------------------------------
    def group_entries(self, btree, heap):
        d = self.d
        if d[heap:heap + 4] != b"HEAP":
            raise ValueError("bad local heap")
        heap_data = struct.unpack_from("<Q", d, heap + 24)[0]
        entries = {}

        def name_at(off):
            s = heap_data + off
            return d[s:d.index(b"\0", s)].decode()

        def walk(node):
            if d[node:node + 4] == b"TREE":
                ntype, level, used = struct.unpack_from("<BBH", d, node + 4)
                if ntype != 0:
                    raise NotImplementedError("non-group B-tree")
                p = node + 8 + 16                       # left/right siblings
                for i in range(used):
                    child = struct.unpack_from("<Q", d, p + 8 + i * 16)[0]
                    walk(child)
            elif d[node:node + 4] == b"SNOD":
                n = struct.unpack_from("<H", d, node + 6)[0]
                p = node + 8
                for i in range(n):
                    name_off, hdr = struct.unpack_from("<QQ", d, p + i * 40)
                    entries[name_at(name_off)] = hdr
            else:
                raise ValueError("bad group node")

        walk(btree)
        return entries

    def root(self):
        if self.root_btree is None:
            for mtype, body, _ in self.messages(self.root_header):
                if mtype == 0x11:                       # symbol table message
                    self.root_btree, self.root_heap = struct.unpack_from("<QQ", self.d, body)
        if self.root_btree is None:
            raise NotImplementedError("new-style (link message) groups")
        return self.group_entries(self.root_btree, self.root_heap)

    # ---- datasets ----------------------------------------------------------------------------
    def read(self, addr):
        d = self.d
        shape, dtype, raw = (), None, None
        for mtype, body, msize in self.messages(addr):
            if mtype == 0x01:                           # dataspace
                ver, rank, flags = struct.unpack_from("<BBB", d, body)
                p = body + (8 if ver == 1 else 4)
                shape = struct.unpack_from("<" + "Q" * rank, d, p) if rank else ()
            elif mtype == 0x03:                         # datatype
                cv, b0, b1, b2, size = struct.unpack_from("<BBBBI", d, body)
                cls = cv & 0x0F
                if cls == 0:
                    dtype = np.dtype(("<" if not b0 & 1 else ">") + ("i" if b0 & 8 else "u") + str(size))
                elif cls == 1:
                    dtype = np.dtype(("<" if not b0 & 1 else ">") + "f" + str(size))
                elif cls == 3:
                    dtype = np.dtype("S" + str(size))
                else:
                    raise NotImplementedError(f"HDF5 datatype class {cls}")
            elif mtype == 0x08:                         # layout
                ver = d[body]
                if ver == 3:
                    lclass = d[body + 1]
                    if lclass == 1:                     # contiguous
                        a, n = struct.unpack_from("<QQ", d, body + 2)
                        raw = None if a == _UNDEF else (a + self.base, n)
                    elif lclass == 0:                   # compact
                        n = struct.unpack_from("<H", d, body + 2)[0]
                        raw = (body + 4, n)
                    else:
                        raise NotImplementedError("chunked HDF5 datasets")
                else:
                    raise NotImplementedError(f"HDF5 layout message version {ver}")
            elif mtype == 0x0B:
                raise NotImplementedError("filtered (compressed) HDF5 datasets")
        if dtype is None:
            raise ValueError("dataset without datatype")
        count = int(np.prod(shape)) if shape else 1
        if raw is None:
            return np.zeros(shape, dtype)
        a, n = raw
        arr = np.frombuffer(d, dtype, count=count, offset=a).reshape(shape)
        if dtype.kind == "S":
            arr = np.char.decode(arr, "utf-8", "replace") if arr.shape else arr.item().decode("utf-8", "replace")
            return arr
        return arr.copy() if arr.shape else arr.item()


def read_hdf5(path, keys=None):
    """-> dict name -> python scalar / numpy array for the datasets of the root group."""
    with open(path, "rb") as f:
        hf = _File(f.read())
    out = {}
    for name, addr in hf.root().items():
        if keys is not None and name not in keys:
            continue
        try:
            out[name] = hf.read(addr)
        except (NotImplementedError, ValueError, struct.error):
            if keys is not None:
                raise                                   # sub-groups / exotic layouts are skipped on a blanket read
    return out
