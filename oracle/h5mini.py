"""ORACLE — TEST INFRASTRUCTURE ONLY (used by oracle/make_golden.py in the build container; nothing here runs on the GPU box).

Minimal HDF5 reader — the image has no h5py.  Enough of the format to read what HDF5.jl / MCMCChainsStorage wrote into the
reference's stored NUTS chain (docs/src/data/subset_simu_single.h5): superblock v0/v1, v1/v2 object headers, symbol-table
groups, link-message groups and "dense" groups whose links sit in ONE fractal-heap direct block, chunked datasets (v1 chunk
B-tree) with the shuffle + deflate filters, contiguous / compact layouts, fixed-size numeric types."""
import struct, zlib, sys
import numpy as np

class H5:
    def __init__(self, path):
        self.raw = open(path, 'rb').read()
        r = self.raw
        assert r[:8] == b"\x89HDF\r\n\x1a\n"
        self.sbver = r[8]
        assert self.sbver in (0, 1), self.sbver
        self.so, self.sl = r[13], r[14]
        assert self.so == 8 and self.sl == 8
        p = 24 if self.sbver == 0 else 28
        base, fsaddr, eof, drv = struct.unpack("<4Q", r[p:p+32]); p += 32
        # root symbol table entry
        self.root = self.sym_entry(p)
    def sym_entry(self, p):
        r = self.raw
        name_off, ohdr, cache, _ = struct.unpack("<QQII", r[p:p+24])
        scratch = r[p+24:p+40]
        return dict(name_off=name_off, ohdr=ohdr, cache=cache, scratch=scratch)
    def read_ohdr(self, addr):
        r = self.raw
        ver = r[addr]
        msgs = []
        if ver == 1:
            nmsg, refc, hsize = struct.unpack("<HII", r[addr+2:addr+12])
            blocks = [(addr+16, hsize)]
            cnt = 0
            while blocks and cnt < nmsg:
                p, sz = blocks.pop(0)
                end = p + sz
                while p + 8 <= end and cnt < nmsg:
                    mtype, msize, flags = struct.unpack("<HHB", r[p:p+5])
                    body = r[p+8:p+8+msize]
                    p += 8 + msize
                    cnt += 1
                    if mtype == 0x10:
                        caddr, clen = struct.unpack("<QQ", body[:16])
                        blocks.append((caddr, clen))
                    else:
                        msgs.append((mtype, body))
        else:
            assert r[addr:addr+4] == b"OHDR", (addr, r[addr:addr+8])
            flags = r[addr+5]
            p = addr + 6
            if flags & 0x20: p += 16
            if flags & 0x10: p += 4
            szb = 1 << (flags & 3)
            csize = int.from_bytes(r[p:p+szb], 'little'); p += szb
            blocks = [(p, csize)]
            track = bool(flags & 4)
            while blocks:
                p, sz = blocks.pop(0)
                end = p + sz
                while p + 4 <= end:
                    mtype = r[p]; msize = struct.unpack("<H", r[p+1:p+3])[0]; mflags = r[p+3]; p += 4
                    if track: p += 2
                    body = r[p:p+msize]; p += msize
                    if mtype == 0x10:
                        caddr, clen = struct.unpack("<QQ", body[:16])
                        assert r[caddr:caddr+4] == b"OCHK"
                        blocks.append((caddr+4, clen-8))
                    elif mtype != 0:
                        msgs.append((mtype, body))
        return msgs
    # ---- groups
    def heap_data(self, addr):
        r = self.raw
        assert r[addr:addr+4] == b"HEAP"
        dsize, free, daddr = struct.unpack("<QQQ", r[addr+8:addr+32])
        return daddr
    def group_entries(self, btree, heap):
        r = self.raw
        hd = self.heap_data(heap)
        out = {}
        def walk(a):
            assert r[a:a+4] == b"TREE", a
            ntype, level, nent = struct.unpack("<BBH", r[a+4:a+8])
            assert ntype == 0
            p = a + 24
            p += 8  # key 0
            for _ in range(nent):
                child, = struct.unpack("<Q", r[p:p+8]); p += 16
                if level > 0: walk(child)
                else:
                    assert r[child:child+4] == b"SNOD"
                    n, = struct.unpack("<H", r[child+6:child+8])
                    q = child + 8
                    for i in range(n):
                        e = self.sym_entry(q); q += 40
                        s = hd + e['name_off']
                        name = r[s:r.index(b"\0", s)].decode()
                        out[name] = e
        walk(btree)
        return out
    def children(self, ohdr):
        msgs = self.read_ohdr(ohdr)
        for t, b in msgs:
            if t == 0x11:
                bt, hp = struct.unpack("<QQ", b[:16])
                return self.group_entries(bt, hp)
        # new-style link messages
        out = {}
        for t, b in msgs:
            if t == 0x6:
                ver, fl = b[0], b[1]; p = 2
                ltype = 0
                if fl & 8: ltype = b[p]; p += 1
                if fl & 4: p += 8
                if fl & 16: p += 1
                ls = 1 << (fl & 3)
                ln = int.from_bytes(b[p:p+ls], 'little'); p += ls
                name = b[p:p+ln].decode(); p += ln
                if ltype == 0:
                    out[name] = dict(ohdr=struct.unpack("<Q", b[p:p+8])[0])
        if out:
            return out
        # "dense" link storage (Link Info message, fractal heap): the links of a small group are the managed objects of the
        # heap's single direct block, stored back to back
        for t, b in msgs:
            if t == 0x2:
                fl = b[1]; p = 2 + (8 if fl & 1 else 0)
                fheap, = struct.unpack("<Q", b[p:p + 8])
                if fheap == 0xffffffffffffffff:
                    return {}
                return self.dense_links(fheap)
        return None
    def dense_links(self, fheap):
        r = self.raw
        assert r[fheap:fheap + 4] == b"FRHP"
        # the root block address sits 8 + 2 + 2 + 1 + 4 + 8*12 + 2 + 8 + 8 + 2 + 2 = bytes into the header: search instead for
        # the direct block that names this heap as its owner (signature, version, heap header address)
        key = b"FHDB\x00" + struct.pack("<Q", fheap)
        a = r.find(key)
        assert a >= 0, "fractal heap without a direct block (indirect root blocks are not supported)"
        p = r.find(b"\x01", a + len(key) + 8)      # first link message: version 1
        out = {}
        while r[p] == 1:
            fl = r[p + 1]; q = p + 2
            ltype = 0
            if fl & 8: ltype = r[q]; q += 1
            if fl & 4: q += 8
            if fl & 16: q += 1
            ls = 1 << (fl & 3)
            ln = int.from_bytes(r[q:q + ls], 'little'); q += ls
            name = r[q:q + ln].decode(); q += ln
            assert ltype == 0
            out[name] = dict(ohdr=struct.unpack("<Q", r[q:q + 8])[0]); p = q + 8
        return out
    # ---- datatypes
    def parse_dtype(self, b):
        cv = b[0]; cls = cv & 15; bits = b[1:4]; size, = struct.unpack("<I", b[4:8])
        if cls == 0:
            signed = bits[0] & 8
            return np.dtype(('<i' if signed else '<u') + str(size)), None
        if cls == 1:
            return np.dtype('<f%d' % size), None
        if cls == 3:
            return np.dtype('S%d' % size), None
        if cls == 9:
            base, _ = self.parse_dtype(b[8:])
            typ = bits[0] & 15
            return np.dtype('V%d' % size), ('vlen', typ, base)
        if cls == 7:
            return np.dtype('V%d' % size), ('ref',)
        if cls == 6:
            return np.dtype('V%d' % size), ('compound',)
        raise NotImplementedError(cls)
    def parse_space(self, b):
        ver = b[0]; rank = b[1]; fl = b[2]
        p = 8 if ver == 1 else 4
        dims = struct.unpack("<%dQ" % rank, b[p:p+8*rank])
        return dims
    def gheap_obj(self, addr, idx):
        r = self.raw
        assert r[addr:addr+4] == b"GCOL", addr
        csize, = struct.unpack("<Q", r[addr+8:addr+16])
        p = addr + 16
        while p < addr + csize:
            i, refc, _, osz = struct.unpack("<HHIQ", r[p:p+16])
            if i == 0: break
            if i == idx: return r[p+16:p+16+osz]
            p += 16 + ((osz + 7) // 8) * 8
        raise KeyError(idx)
    def decode_vlen(self, arr, info):
        out = []
        for v in arr.reshape(-1):
            n, addr, idx = struct.unpack("<IQI", v.tobytes())
            data = self.gheap_obj(addr, idx) if n else b""
            if info[1] == 1: out.append(data[:n].decode())
            else: out.append(np.frombuffer(data, dtype=info[2], count=n))
        return out
    def read_dataset_msgs(self, msgs):
        dt = info = dims = layout = None; filters = []
        for t, b in msgs:
            if t == 1: dims = self.parse_space(b)
            elif t == 3: dt, info = self.parse_dtype(b)
            elif t == 8: layout = b
            elif t == 0xB:
                ver = b[0]; nf = b[1]
                p = 8 if ver == 1 else 2
                for _ in range(nf):
                    fid, = struct.unpack("<H", b[p:p+2]); p += 2
                    if ver == 1 or fid >= 256:
                        nlen, = struct.unpack("<H", b[p:p+2]); p += 2
                    else: nlen = 0
                    fl, ncv = struct.unpack("<HH", b[p:p+4]); p += 4
                    if ver == 1: nlen = (nlen + 7)//8*8
                    p += nlen
                    cv = struct.unpack("<%dI" % ncv, b[p:p+4*ncv]); p += 4*ncv
                    if ver == 1 and ncv % 2: p += 4
                    filters.append((fid, cv))
        return dt, info, dims, layout, filters
    def read_data(self, ohdr):
        r = self.raw
        dt, info, dims, layout, filters = self.read_dataset_msgs(self.read_ohdr(ohdr))
        if dt is None: return None
        n = int(np.prod(dims)) if dims else 1
        ver = layout[0]
        assert ver == 3, ver
        cls = layout[1]
        if cls == 0:
            sz, = struct.unpack("<H", layout[2:4]); buf = layout[4:4+sz]
            arr = np.frombuffer(buf, dtype=dt, count=n)
        elif cls == 1:
            addr, sz = struct.unpack("<QQ", layout[2:18])
            arr = np.frombuffer(r[addr:addr+sz], dtype=dt, count=n) if addr != 0xffffffffffffffff else np.zeros(n, dt)
        else:
            rank = layout[2]; bt, = struct.unpack("<Q", layout[3:11])
            cd = struct.unpack("<%dI" % rank, layout[11:11+4*rank])
            cdims = cd[:-1]
            arr = np.zeros(dims, dtype=dt)
            def walk(a):
                assert r[a:a+4] == b"TREE"
                ntype, level, nent = struct.unpack("<BBH", r[a+4:a+8])
                p = a + 24
                for _ in range(nent):
                    csz, fmask = struct.unpack("<II", r[p:p+8])
                    offs = struct.unpack("<%dQ" % rank, r[p+8:p+8+8*rank]); p += 8 + 8*rank
                    child, = struct.unpack("<Q", r[p:p+8]); p += 8
                    if level > 0: walk(child); continue
                    buf = r[child:child+csz]
                    for k, (fid, cv) in reversed(list(enumerate(filters))):
                        if fmask & (1 << k): continue
                        if fid == 1: buf = zlib.decompress(buf)
                        elif fid == 2:
                            es = cv[0]; m = len(buf)//es
                            buf = np.frombuffer(buf[:m*es], np.uint8).reshape(es, m).T.tobytes() + buf[m*es:]
                        elif fid == 3: buf = buf[:-4]
                        else: raise NotImplementedError(fid)
                    blk = np.frombuffer(buf, dtype=dt, count=int(np.prod(cdims))).reshape(cdims)
                    sl = tuple(slice(o, min(o+c, d)) for o, c, d in zip(offs, cdims, dims))
                    arr[sl] = blk[tuple(slice(0, s.stop - s.start) for s in sl)]
            if bt != 0xffffffffffffffff: walk(bt)
            return arr if info is None else (self.decode_vlen(arr, info) if info[0]=='vlen' else arr)
        arr = arr.reshape(dims) if dims else arr
        if info and info[0] == 'vlen': return self.decode_vlen(arr, info)
        return arr
    def attrs(self, ohdr):
        out = {}
        for t, b in self.read_ohdr(ohdr):
            if t != 0xC: continue
            ver = b[0]
            nsz, dsz, ssz = struct.unpack("<HHH", b[2:8])
            p = 8
            if ver == 3: p += 1
            pad = (lambda x: (x+7)//8*8) if ver == 1 else (lambda x: x)
            name = b[p:p+nsz].split(b"\0")[0].decode(); p += pad(nsz)
            dt, info = self.parse_dtype(b[p:p+dsz]); p += pad(dsz)
            dims = self.parse_space(b[p:p+ssz]) if b[p+1] else (); p += pad(ssz)
            n = int(np.prod(dims)) if dims else 1
            arr = np.frombuffer(b[p:p+n*dt.itemsize], dtype=dt, count=n)
            if info and info[0] == 'vlen': arr = self.decode_vlen(arr, info)
            out[name] = arr
        return out
    def tree(self, ohdr=None, prefix="", out=None):
        if ohdr is None: ohdr = self.root['ohdr']
        if out is None: out = {}
        ch = self.children(ohdr)
        if ch is None:
            out[prefix] = ohdr; return out
        out[prefix + "/"] = ohdr
        for k, e in ch.items():
            self.tree(e['ohdr'], prefix + "/" + k, out)
        return out

if __name__ == "__main__":
    f = H5(sys.argv[1])
    for k, oh in f.tree().items():
        if k.endswith("/"):
            print(k, "attrs:", {a: (v if len(v) < 8 else f"<{len(v)}>") for a, v in f.attrs(oh).items()})
        else:
            dt, info, dims, layout, filters = f.read_dataset_msgs(f.read_ohdr(oh))
            print(k, dt, info, dims, filters, {a: v for a, v in f.attrs(oh).items()})
