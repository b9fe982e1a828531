"""Minimal LMDB (data.mdb, format version 1) reader and bulk writer in pure Python.

The reference stores its datasets with py-lmdb (build_lmdb.py:72-112, imagereader.py:103-144); neither py-lmdb nor
liblmdb exists in this image, so the on-disk B+tree is handled here directly:

  page (4096 B): header {pgno u64, pad u16, flags u16, lower u16, upper u16 | overflow: pages u32}, then u16 node
                 offsets growing up and nodes growing down;
  meta pages 0/1: header + {magic 0xBEEFC0DE, version 1, address, mapsize, dbs[FREE, MAIN] (48 B each: pad/psize u32,
                 flags u16, depth u16, branch, leaf, overflow pages, entries, root u64), last_pg, txnid};
  node: {lo u16, hi u16, flags u16, ksize u16, key, data}; leaf data size = lo | hi << 16, F_BIGDATA (0x01) data is
                 the u64 page number of an overflow run; branch child page = lo | hi << 16 | flags << 32.

Reader: read-only, memory-mapped, ordered iteration and point lookups.  Writer: builds a fresh environment in one
streaming pass over the keys in LMDB's default byte order: overflow runs (values above the node limit) and leaf pages
go to the file as they fill up, only (first key, page number) per leaf stays in memory (the reference commits every
1000 records into a 5 TB map, build_lmdb.py:80-100: a dataset never has to fit in RAM).
PARITY UNPINNED: no liblmdb here to cross-check against and the reference ships no data.mdb; the writer follows the
published format above and is verified by round trips through the reader (tests/test_cpu_dataplane.py).
"""
import mmap
import os
import struct

PAGE = 4096
HDR = 16
P_BRANCH, P_LEAF, P_OVERFLOW, P_META = 0x01, 0x02, 0x04, 0x08
F_BIGDATA = 0x01
MAGIC = 0xBEEFC0DE
INVALID = 0xFFFFFFFFFFFFFFFF
NODE_MAX = (((PAGE - HDR) // 2) & ~1) - 2   # mdb: me_nodemax = (((psize - PAGEHDRSZ) / MDB_MINKEYS) & -2) - sizeof(indx_t) = 2038


class LmdbError(Exception):
    pass


class Environment:
    """Read-only view of <path>/data.mdb (or of the file itself)."""

    def __init__(self, path):
        f = os.path.join(path, 'data.mdb') if os.path.isdir(path) else path
        if not os.path.exists(f):
            raise LmdbError('no LMDB environment at %s' % path)
        self._fh = open(f, 'rb')
        self._mm = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        metas = []
        for pg in (0, 1):
            off = pg * PAGE + HDR
            magic, version = struct.unpack_from('<II', self._mm, off)
            if magic != MAGIC or version != 1:
                continue
            psize = struct.unpack_from('<I', self._mm, off + 24)[0]            # dbs[FREE].md_pad
            main = struct.unpack_from('<IHHQQQQQ', self._mm, off + 24 + 48)   # dbs[MAIN]
            last_pg, txnid = struct.unpack_from('<QQ', self._mm, off + 24 + 96)
            metas.append((txnid, psize, main, last_pg))
        if not metas:
            raise LmdbError('not an LMDB data file: %s' % f)
        txnid, self.psize, main, self.last_pg = max(metas, key=lambda m: m[0])
        if self.psize != PAGE:
            raise LmdbError('page size %d unsupported' % self.psize)
        _, _, self.depth, _, _, _, self.entries, self.root = main

    def close(self):
        self._mm.close()
        self._fh.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def stat(self):
        return {'entries': self.entries, 'depth': self.depth, 'psize': self.psize}

    # -- page helpers ------------------------------------------------------------------------------------------
    def _page(self, pgno):
        off = pgno * PAGE
        flags, lower = struct.unpack_from('<HH', self._mm, off + 10)
        n = (lower - HDR) // 2
        ptrs = struct.unpack_from('<%dH' % n, self._mm, off + HDR) if n else ()
        return off, flags, ptrs

    def _leaf_node(self, off, p):
        lo, hi, flags, ksize = struct.unpack_from('<HHHH', self._mm, off + p)
        key = bytes(self._mm[off + p + 8: off + p + 8 + ksize])
        dsize = lo | (hi << 16)
        dpos = off + p + 8 + ksize
        if flags & F_BIGDATA:
            ov = struct.unpack_from('<Q', self._mm, dpos)[0]
            start = ov * PAGE + HDR
            return key, (start, dsize)
        return key, (dpos, dsize)

    def _branch_node(self, off, p):
        lo, hi, flags, ksize = struct.unpack_from('<HHHH', self._mm, off + p)
        key = bytes(self._mm[off + p + 8: off + p + 8 + ksize])
        return key, lo | (hi << 16) | (flags << 32)

    def _walk(self, pgno):
        off, flags, ptrs = self._page(pgno)
        if flags & P_LEAF:
            for p in ptrs:
                yield self._leaf_node(off, p)
        elif flags & P_BRANCH:
            for p in ptrs:
                _, child = self._branch_node(off, p)
                yield from self._walk(child)
        else:
            raise LmdbError('unexpected page flags 0x%x at page %d' % (flags, pgno))

    # -- public API (subset of py-lmdb's cursor / txn) -----------------------------------------------------------------
    def keys(self):
        if self.root == INVALID:
            return
        for k, _ in self._walk(self.root):
            yield k

    def items(self):
        if self.root == INVALID:
            return
        for k, (pos, size) in self._walk(self.root):
            yield k, bytes(self._mm[pos:pos + size])

    def get(self, key):
        if self.root == INVALID:
            return None
        key = bytes(key)
        pgno = self.root
        while True:
            off, flags, ptrs = self._page(pgno)
            if flags & P_LEAF:
                lo, hi = 0, len(ptrs) - 1
                while lo <= hi:
                    mid = (lo + hi) // 2
                    k, (pos, size) = self._leaf_node(off, ptrs[mid])
                    if k == key:
                        return bytes(self._mm[pos:pos + size])
                    if k < key:
                        lo = mid + 1
                    else:
                        hi = mid - 1
                return None
            # branch: last child whose separator key <= key (node 0 has the implicit lowest key)
            lo, hi = 1, len(ptrs) - 1
            idx = 0
            while lo <= hi:
                mid = (lo + hi) // 2
                k, _ = self._branch_node(off, ptrs[mid])
                if k <= key:
                    idx = mid
                    lo = mid + 1
                else:
                    hi = mid - 1
            _, pgno = self._branch_node(off, ptrs[idx])


def _page_bytes(pgno, flags, nodes):
    """Assemble a branch/leaf page from already-encoded nodes (in key order)."""
    buf = bytearray(PAGE)
    upper = PAGE
    ptrs = []
    for nd in nodes:
        sz = (len(nd) + 1) & ~1
        upper -= sz
        buf[upper:upper + len(nd)] = nd
        ptrs.append(upper)
    lower = HDR + 2 * len(ptrs)
    assert lower <= upper, 'page overflow'
    struct.pack_into('<QHHHH', buf, 0, pgno, 0, flags, lower, upper)
    struct.pack_into('<%dH' % len(ptrs), buf, HDR, *ptrs)
    return bytes(buf)


def write_environment_stream(path, sorted_keys, value_of, map_size=None):
    """Create <path>/data.mdb (+ an empty lock.mdb).  ``sorted_keys``: unique key bytes in ascending byte order;
    ``value_of(key)`` is called once per key, in that order, and its bytes are written out before the next call --
    memory use is one leaf page plus one value, whatever the size of the dataset."""
    os.makedirs(path, exist_ok=True)
    next_pg = [2]
    counts = dict(leaf=0, branch=0, over=0, entries=0)

    def alloc(n=1):
        p = next_pg[0]
        next_pg[0] += n
        return p

    with open(os.path.join(path, 'data.mdb'), 'wb') as fh:
        def put_page(pg, data):
            fh.seek(pg * PAGE)
            fh.write(data)

        level = []           # (first key, pgno) of every leaf
        cur, cur_size, cur_first = [], HDR, None

        def flush_leaf():
            nonlocal cur, cur_size, cur_first
            if not cur:
                return
            pg = alloc()
            put_page(pg, _page_bytes(pg, P_LEAF, cur))
            level.append((cur_first, pg))
            counts['leaf'] += 1
            cur, cur_size, cur_first = [], HDR, None

        prev = None
        for k in sorted_keys:
            k = bytes(k)
            if prev is not None and k <= prev:
                raise LmdbError('duplicate key %r' % k if k == prev else 'keys not in ascending order at %r' % k)
            prev = k
            if len(k) > 511:
                raise LmdbError('key too long')
            v = bytes(value_of(k))
            counts['entries'] += 1
            if 8 + len(k) + len(v) > NODE_MAX:
                npg = (HDR + len(v) + PAGE - 1) // PAGE
                ov = alloc(npg)
                hdr = bytearray(HDR)
                struct.pack_into('<QHHI', hdr, 0, ov, 0, P_OVERFLOW, npg)
                put_page(ov, bytes(hdr) + v)
                counts['over'] += npg
                node = struct.pack('<HHHH', len(v) & 0xFFFF, len(v) >> 16, F_BIGDATA, len(k)) + k + struct.pack('<Q', ov)
            else:
                node = struct.pack('<HHHH', len(v) & 0xFFFF, len(v) >> 16, 0, len(k)) + k + v
            need = ((len(node) + 1) & ~1) + 2
            if cur_size + need > PAGE:
                flush_leaf()
            if cur_first is None:
                cur_first = k
            cur.append(node)
            cur_size += need
        flush_leaf()

        depth = 1 if level else 0
        while len(level) > 1:
            nxt = []
            cur, cur_size, cur_first = [], HDR, None
            for k, pg in level:
                key = b'' if not cur else k          # node 0 of a branch page carries no key
                node = struct.pack('<HHHH', pg & 0xFFFF, (pg >> 16) & 0xFFFF, (pg >> 32) & 0xFFFF, len(key)) + key
                need = ((len(node) + 1) & ~1) + 2
                if cur and cur_size + need > PAGE:
                    bp = alloc()
                    put_page(bp, _page_bytes(bp, P_BRANCH, cur))
                    nxt.append((cur_first, bp))
                    counts['branch'] += 1
                    cur, cur_size, cur_first = [], HDR, None
                    node = struct.pack('<HHHH', pg & 0xFFFF, (pg >> 16) & 0xFFFF, (pg >> 32) & 0xFFFF, 0)
                    need = ((len(node) + 1) & ~1) + 2
                if cur_first is None:
                    cur_first = k
                cur.append(node)
                cur_size += need
            if cur:
                bp = alloc()
                put_page(bp, _page_bytes(bp, P_BRANCH, cur))
                nxt.append((cur_first, bp))
                counts['branch'] += 1
            level = nxt
            depth += 1
        root = level[0][1] if level else INVALID
        last_pg = next_pg[0] - 1
        if map_size is None:
            map_size = max(1 << 20, (last_pg + 1) * PAGE)

        def meta(pgno, txnid):
            buf = bytearray(PAGE)
            struct.pack_into('<QHHHH', buf, 0, pgno, 0, P_META, 0, 0)
            off = HDR
            struct.pack_into('<IIQQ', buf, off, MAGIC, 1, 0, map_size)
            struct.pack_into('<IHHQQQQQ', buf, off + 24, PAGE, 0, 0, 0, 0, 0, 0, INVALID)                  # FREE_DBI
            struct.pack_into('<IHHQQQQQ', buf, off + 72, 0, 0, depth, counts['branch'], counts['leaf'], counts['over'], counts['entries'], root)   # MAIN_DBI
            struct.pack_into('<QQ', buf, off + 120, max(last_pg, 1), txnid)
            return bytes(buf)

        put_page(0, meta(0, 0))
        put_page(1, meta(1, 1))
        fh.truncate((last_pg + 1) * PAGE)
    open(os.path.join(path, 'lock.mdb'), 'ab').close()
    return counts['entries']


def write_environment(path, items, map_size=None):
    """``items``: iterable of (key bytes, value bytes) held in memory (tests, small sets); keys must be unique and are
    stored in LMDB's default (bytewise) order.  Large datasets: write_environment_stream."""
    table = {}
    for k, v in items:
        k = bytes(k)
        if k in table:
            raise LmdbError('duplicate key %r' % k)
        table[k] = v
    return write_environment_stream(path, sorted(table), table.__getitem__, map_size)
