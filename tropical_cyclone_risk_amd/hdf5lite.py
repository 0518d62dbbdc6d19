"""A minimal read-only HDF5 / NetCDF-4 reader: enough of the HDF5 file format to read
the variables and attributes of the files the reference ships and writes (`intensity/data/*.nc`,
`thermo_*.nc`, `env_wnd_*.nc`, `land/<B>.nc`) where neither xarray, netCDF4 nor h5py exists — which
includes the GPU box image.

Supported (HDF5 File Format Specification 3.0): superblock 0-3; object headers v1 and v2 with
continuation blocks; groups through link messages (compact) or a v1 symbol table (B-tree + local
heap); dataspace v1/v2; fixed-point, floating-point and fixed-length string datatypes; contiguous,
compact and chunked (v1 B-tree index) layouts v3; deflate, shuffle and fletcher32 filters; fill
values; attributes v1-v3 of those datatypes.  Not supported, and reported as such: dense group /
attribute storage (fractal heaps), layout v4 chunk indices, variable-length and compound types
(NetCDF-4's DIMENSION_LIST / REFERENCE_LIST bookkeeping attributes are skipped, not needed to read data).

    f = File(path);  f.keys();  f['land'] -> ndarray;  f.attrs('time') -> {'units': ..., 'calendar': ...}
"""
import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class Unsupported(RuntimeError):
    pass


class File:
    def __init__(self, path):
        with open(path, 'rb') as f:
            self.b = f.read()
        self.path = path
        self._superblock()
        self._objects = {}
        self._walk_group(self.root_addr, '')

    # ------------------------------------------------------------------ low level
    def _u(self, off, n):
        return int.from_bytes(self.b[off:off + n], 'little')

    def _superblock(self):
        b = self.b
        pos = 0
        while b[pos:pos + 8] != b'\x89HDF\r\n\x1a\n':      # may sit at 0, 512, 1024, ...
            pos = 512 if pos == 0 else pos * 2
            if pos >= len(b):
                raise Unsupported('%s: no HDF5 signature' % self.path)
        ver = b[pos + 8]
        if ver in (0, 1):
            self.so, self.sl = b[pos + 13], b[pos + 14]
            p = pos + 24 + (4 if ver == 1 else 0)
            self.base = self._u(p, self.so)
            p += 4 * self.so                                   # base, free-space, eof, driver
            self.root_addr = self._u(p + self.so, self.so)     # symbol table entry: link name offset, object header address
        elif ver in (2, 3):
            self.so, self.sl = b[pos + 9], b[pos + 10]
            p = pos + 12
            self.base = self._u(p, self.so)
            self.root_addr = self._u(p + 3 * self.so, self.so)
        else:
            raise Unsupported('superblock version %d' % ver)
        if self.so != 8 or self.sl != 8:
            raise Unsupported('offset/length sizes %d/%d' % (self.so, self.sl))

    # ------------------------------------------------------------------ object headers
    def _messages(self, addr):
        """Yield (type, flags, body bytes) of every message of the object header at addr."""
        b = self.b
        addr += self.base
        if b[addr:addr + 4] == b'OHDR':
            flags = b[addr + 5]
            p = addr + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            nsz = 1 << (flags & 3)
            chunk = self._u(p, nsz)
            p += nsz
            blocks = [(p, p + chunk)]
            order = 2 if flags & 4 else 0
            while blocks:
                p, end = blocks.pop(0)
                while p + 4 + order <= end:
                    mtype, msize, mflags = b[p], self._u(p + 1, 2), b[p + 3]
                    q = p + 4 + order
                    body = b[q:q + msize]
                    if mtype == 0x10:
                        caddr, clen = self._u(q, 8) + self.base, self._u(q + 8, 8)
                        if b[caddr:caddr + 4] != b'OCHK':
                            raise Unsupported('bad continuation block')
                        blocks.append((caddr + 4, caddr + clen - 4))
                    elif mtype != 0:
                        yield mtype, mflags, body
                    p = q + msize
        else:                                                   # version 1
            if b[addr] != 1:
                raise Unsupported('object header version %d' % b[addr])
            nmsg = self._u(addr + 2, 2)
            size = self._u(addr + 8, 4)
            blocks = [(addr + 16, addr + 16 + size)]
            seen = 0
            while blocks and seen < nmsg:
                p, end = blocks.pop(0)
                while p + 8 <= end and seen < nmsg:
                    mtype, msize, mflags = self._u(p, 2), self._u(p + 2, 2), b[p + 4]
                    body = b[p + 8:p + 8 + msize]
                    seen += 1
                    if mtype == 0x10:
                        blocks.append((self._u(p + 8, 8) + self.base, self._u(p + 8, 8) + self.base + self._u(p + 16, 8)))
                    elif mtype != 0:
                        yield mtype, mflags, body
                    p += 8 + msize

    # ------------------------------------------------------------------ groups
    def _walk_group(self, addr, prefix):
        for mtype, _, body in self._messages(addr):
            if mtype == 0x06:                                   # link
                name, target = self._link(body)
                if target is not None:
                    self._register(prefix + name, target)
            elif mtype == 0x11:                                 # symbol table: v1 B-tree + local heap
                self._symbol_table(self._u_b(body, 0, 8), self._u_b(body, 8, 8), prefix)
            elif mtype == 0x02:                                 # link info: dense storage if a fractal heap is set
                flags = body[1]
                p = 2 + (8 if flags & 1 else 0)
                if self._u_b(body, p, 8) != UNDEF:
                    raise Unsupported('%s: dense link storage (fractal heap)' % (prefix or '/'))

    @staticmethod
    def _u_b(body, off, n):
        return int.from_bytes(body[off:off + n], 'little')

    def _link(self, body):
        flags = body[1]
        p = 2
        ltype = 0
        if flags & 0x08:
            ltype = body[p]; p += 1
        if flags & 0x04:
            p += 8
        if flags & 0x10:
            p += 1
        nlen_size = 1 << (flags & 3)
        nlen = self._u_b(body, p, nlen_size); p += nlen_size
        name = body[p:p + nlen].decode('utf-8'); p += nlen
        return name, (self._u_b(body, p, 8) if ltype == 0 else None)

    def _symbol_table(self, btree, heap, prefix):
        b = self.b
        heap += self.base
        if b[heap:heap + 4] != b'HEAP':
            raise Unsupported('bad local heap')
        data = self._u(heap + 8 + 2 * self.sl, self.so) + self.base

        def node(addr):
            addr += self.base
            if b[addr:addr + 4] == b'TREE':
                level, used = b[addr + 5], self._u(addr + 6, 2)
                p = addr + 8 + 2 * self.so + self.sl            # first key, then child pointers interleaved
                for _ in range(used):
                    child = self._u(p, self.so)
                    node(child)
                    p += self.so + self.sl
            elif b[addr:addr + 4] == b'SNOD':
                n = self._u(addr + 6, 2)
                p = addr + 8
                for _ in range(n):
                    noff, oaddr = self._u(p, self.so), self._u(p + self.so, self.so)
                    q = data + noff
                    name = b[q:b.index(b'\0', q)].decode('utf-8')
                    self._register(prefix + name, oaddr)
                    p += 2 * self.so + 8 + 16
        node(btree)

    def _register(self, name, addr):
        types = [m for m, _, _ in self._messages(addr)]
        if 0x08 in types:                                       # has a data layout: a dataset
            self._objects[name] = addr
        else:
            self._walk_group(addr, name + '/')

    # ------------------------------------------------------------------ datasets
    def keys(self):
        return sorted(self._objects)

    def __contains__(self, k):
        return k in self._objects

    @staticmethod
    def _dtype(body):
        cls, ver = body[0] & 0x0F, body[0] >> 4
        bits0 = body[1]
        size = int.from_bytes(body[4:8], 'little')
        order = '>' if bits0 & 1 else '<'
        if cls == 0:
            return np.dtype('%s%s%d' % (order, 'i' if bits0 & 8 else 'u', size))
        if cls == 1:
            return np.dtype('%sf%d' % (order, size))
        if cls == 3:
            return np.dtype('S%d' % size)
        raise Unsupported('datatype class %d (version %d)' % (cls, ver))

    @staticmethod
    def _dataspace(body):
        ver, rank, flags = body[0], body[1], body[2]
        if ver == 1:
            p = 8
        elif ver == 2:
            if body[3] == 2:
                return None                                     # null dataspace
            p = 4
        else:
            raise Unsupported('dataspace version %d' % ver)
        return tuple(int.from_bytes(body[p + 8 * i:p + 8 * i + 8], 'little') for i in range(rank))

    def _info(self, name):
        dt = shape = layout = fill = None
        filters = []
        attrs = {}
        for mtype, _, body in self._messages(self._objects[name]):
            if mtype == 0x03:
                dt = self._dtype(body)
            elif mtype == 0x01:
                shape = self._dataspace(body)
            elif mtype == 0x08:
                layout = body
            elif mtype == 0x0B:
                filters = self._filters(body)
            elif mtype == 0x05:
                fill = body
            elif mtype == 0x0C:
                try:
                    k, v = self._attribute(body)
                    attrs[k] = v
                except Unsupported:
                    pass                                        # DIMENSION_LIST & co.
            elif mtype == 0x15:
                flags = body[1]
                p = 2 + (2 if flags & 1 else 0)
                if self._u_b(body, p, 8) != UNDEF:
                    raise Unsupported('%s: dense attribute storage (fractal heap)' % name)
        return dt, shape, layout, filters, fill, attrs

    @staticmethod
    def _filters(body):
        ver, n = body[0], body[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid = int.from_bytes(body[p:p + 2], 'little')
            if ver == 1 or fid >= 256:
                nlen = int.from_bytes(body[p + 2:p + 4], 'little'); q = p + 4
            else:
                nlen = 0; q = p + 2
            ncv = int.from_bytes(body[q + 2:q + 4], 'little')
            q += 4
            if nlen:
                q += (nlen + 7) // 8 * 8 if ver == 1 else nlen
            cv = [int.from_bytes(body[q + 4 * i:q + 4 * i + 4], 'little') for i in range(ncv)]
            q += 4 * ncv
            if ver == 1 and ncv % 2:
                q += 4
            out.append((fid, cv))
            p = q
        return out

    def _attribute(self, body):
        ver = body[0]
        nsz, tsz, ssz = (int.from_bytes(body[2 + 2 * i:4 + 2 * i], 'little') for i in range(3))
        p = 8 if ver == 1 else (8 if ver == 2 else 9)
        pad = (lambda n: (n + 7) // 8 * 8) if ver == 1 else (lambda n: n)
        name = body[p:p + nsz].split(b'\0')[0].decode('utf-8'); p += pad(nsz)
        dt = self._dtype(body[p:p + tsz]); p += pad(tsz)
        shape = self._dataspace(body[p:p + ssz]); p += pad(ssz)
        n = int(np.prod(shape)) if shape else 1
        raw = np.frombuffer(body[p:p + n * dt.itemsize], dtype=dt, count=n)
        if dt.kind == 'S':
            v = raw[0].split(b'\0')[0].decode('utf-8', 'replace') if n == 1 else [x.decode('utf-8', 'replace') for x in raw]
        else:
            v = raw[0].item() if (shape is None or shape == ()) or n == 1 else np.array(raw)
        return name, v

    def attrs(self, name):
        return self._info(name)[5]

    def __getitem__(self, name):
        dt, shape, layout, filters, fill, _ = self._info(name)
        if dt is None or shape is None or layout is None:
            raise Unsupported('%s: incomplete dataset header' % name)
        n = int(np.prod(shape)) if shape else 1
        ver, cls = layout[0], layout[1]
        if ver != 3:
            raise Unsupported('%s: data layout version %d' % (name, ver))
        if cls == 0:                                            # compact
            size = self._u_b(layout, 2, 2)
            return np.frombuffer(layout[4:4 + size], dtype=dt, count=n).reshape(shape).copy()
        if cls == 1:                                            # contiguous
            addr = self._u_b(layout, 2, 8)
            if addr == UNDEF:
                return self._filled(dt, shape, fill)
            addr += self.base
            return np.frombuffer(self.b[addr:addr + n * dt.itemsize], dtype=dt, count=n).reshape(shape).copy()
        if cls != 2:
            raise Unsupported('%s: layout class %d' % (name, cls))
        rank = layout[2] - 1
        btree = self._u_b(layout, 3, 8)
        cdims = tuple(self._u_b(layout, 11 + 4 * i, 4) for i in range(rank))
        out = self._filled(dt, shape, fill)
        if btree == UNDEF:
            return out
        csize = int(np.prod(cdims)) * dt.itemsize
        for offs, fmask, addr, nbytes in self._chunks(btree, rank):
            raw = self.b[addr + self.base:addr + self.base + nbytes]
            for i, (fid, cv) in reversed(list(enumerate(filters))):
                if fmask & (1 << i):
                    continue
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:
                    es = cv[0] if cv else dt.itemsize
                    a = np.frombuffer(raw, dtype=np.uint8)
                    m = len(a) // es
                    raw = a[:m * es].reshape(es, m).T.tobytes() + a[m * es:].tobytes()
                elif fid == 3:
                    raw = raw[:-4]
                else:
                    raise Unsupported('%s: filter %d' % (name, fid))
            chunk = np.frombuffer(raw[:csize], dtype=dt).reshape(cdims)
            sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, shape))
            out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]
        return out

    @staticmethod
    def _filled(dt, shape, fill):
        out = np.zeros(shape, dtype=dt.newbyteorder('='))
        if fill is not None:
            ver = fill[0]
            try:
                if ver in (1, 2):
                    defined = fill[3] if ver == 2 else 1
                    if defined:
                        size = int.from_bytes(fill[4:8], 'little')
                        if size == dt.itemsize:
                            out[...] = np.frombuffer(fill[8:8 + size], dtype=dt)[0]
                elif ver == 3 and fill[1] & 0x20:
                    size = int.from_bytes(fill[2:6], 'little')
                    if size == dt.itemsize:
                        out[...] = np.frombuffer(fill[6:6 + size], dtype=dt)[0]
            except Exception:
                pass
        return out

    def _chunks(self, addr, rank):
        b = self.b
        a = addr + self.base
        if b[a:a + 4] != b'TREE' or b[a + 4] != 1:
            raise Unsupported('chunk index is not a v1 B-tree')
        level, used = b[a + 5], self._u(a + 6, 2)
        p = a + 8 + 2 * self.so
        ksz = 8 + 8 * (rank + 1)
        for _ in range(used):
            nbytes, fmask = self._u(p, 4), self._u(p + 4, 4)
            offs = tuple(self._u(p + 8 + 8 * i, 8) for i in range(rank))
            child = self._u(p + ksz, self.so)
            if level == 0:
                yield offs, fmask, child, nbytes
            else:
                yield from self._chunks(child, rank)
            p += ksz + self.so


def read_variables(path):
    """{name: (array with _FillValue/missing_value -> NaN and scale/offset applied for floats, attrs)}."""
    f = File(path)
    out = {}
    for k in f.keys():
        try:
            a, at = f[k], f.attrs(k)
        except Unsupported:
            continue
        if a.dtype.kind == 'f':
            a = a.astype(np.float64)
            for key in ('_FillValue', 'missing_value'):
                if key in at and np.isscalar(at[key]):
                    a[a == np.float64(at[key])] = np.nan
        if 'scale_factor' in at or 'add_offset' in at:
            a = a.astype(np.float64) * float(at.get('scale_factor', 1.0)) + float(at.get('add_offset', 0.0))
        out[k] = (a, at)
    return out
