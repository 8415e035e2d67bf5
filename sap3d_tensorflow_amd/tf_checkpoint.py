"""Reader and writer for the checkpoint files the reference's trainers produce and consume: TensorFlow's "tensor bundle"
(checkpoint format V2, the default of tf.train.Saver in every TF 1.x release the reference can run on).

Reference call sites: `saver = tf.train.Saver(var_list=..., max_to_keep=10)` (train.py:180-185),
`saver.save(sess, './model/<run>/p3d_<step>.ckpt')` (train.py:266-267), `tf.train.get_checkpoint_state(dir)` +
`saver.restore(sess, ckpt.model_checkpoint_path)` (train.py:204-210, gen_pred.py:57-64).  The variable names stored in
the bundle are the graph's variable names, which are exactly the parameter names of libp3dhip,
so `P3DSession.load(read_checkpoint(prefix))` restores a model trained by the reference and `write_checkpoint(prefix,
sess.save())` produces one the reference can restore.

PARITY UNPINNED: TensorFlow is not installed in this image and the reference ships no checkpoint, so no file written
by TensorFlow was available; the format is restated from TensorFlow's published definitions --
tensorflow/core/protobuf/tensor_bundle.proto (BundleHeaderProto, BundleEntryProto), tensorflow/core/lib/io/format.h +
table_builder.cc (the LevelDB-style table of `<prefix>.index`: prefix-compressed blocks with restart arrays, 5-byte
block trailers, 48-byte footer with magic 0xdb4775248b80fb57) and tensorflow/core/lib/hash/crc32c.h (masked CRC-32C).
What pins it here: CRC-32C known answers, wire-format checks, a two-variable `.index` file assembled by hand from those
definitions and compared byte for byte, and write -> read round trips (tests/test_tf_checkpoint.py).

A bundle is `<prefix>.index` (the table: key "" -> header, tensor name -> entry {dtype, shape, shard, offset, size,
crc32c}) plus `<prefix>.data-0000N-of-0000M` (raw little-endian tensor bytes).  Only what tf.train.Saver writes for
this model is supported: DT_FLOAT (and DT_INT32 / DT_INT64 scalars such as global_step), unpartitioned variables,
uncompressed index blocks.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
_MASK_DELTA = 0xA282EAD8
DT_FLOAT, DT_INT32, DT_INT64 = 1, 3, 9
_DTYPES = {DT_FLOAT: np.dtype("<f4"), DT_INT32: np.dtype("<i4"), DT_INT64: np.dtype("<i8")}
_DT_OF = {np.dtype("float32"): DT_FLOAT, np.dtype("int32"): DT_INT32, np.dtype("int64"): DT_INT64}


# ---- CRC-32C (Castagnoli), masked as LevelDB / TensorFlow store it -------------------------------------------------------
def _make_table():
    poly = 0x82F63B78
    t = np.zeros(256, np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (poly if c & 1 else 0)
        t[i] = c
    return t


_TABLE = _make_table()
_TABLE_LIST = [int(v) for v in _TABLE]


def crc32c(data, crc=0):
    """CRC-32C of bytes-like `data` (software, byte at a time in Python for small inputs; large tensors go through the
    native routine of libp3dhip when the library is loaded)."""
    mv = memoryview(data).cast("B")
    if len(mv) >= 1 << 16:
        native = _native_crc()
        if native is not None:
            return native(mv, crc)
    if len(mv) >= 1 << 12:
        return _crc32c_numpy(mv, crc)
    c = crc ^ 0xFFFFFFFF
    tab = _TABLE_LIST
    for b in mv:
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


_TABLES8 = None


def _crc32c_numpy(mv, crc):
    """Slicing-by-8 with numpy for when libp3dhip cannot be loaded: the buffer is cut into 256 lanes that are checksummed
    side by side (vectorised over the lanes, eight bytes per step through eight tables) and stitched together with the
    zero-extension operator of the CRC (crc(A || B) = shift(crc(A), len(B)) xor crc'(B)); the ragged tail goes byte by byte."""
    global _TABLES8
    if _TABLES8 is None:
        t = np.zeros((8, 256), np.uint32)
        t[0] = _TABLE
        for k in range(1, 8):
            t[k] = t[0][t[k - 1] & 0xFF] ^ (t[k - 1] >> 8)
        _TABLES8 = t
    t = _TABLES8
    data = np.frombuffer(mv, np.uint8)
    lanes = 256
    per = (len(data) // lanes) // 8 * 8
    c = crc ^ 0xFFFFFFFF
    if per:
        body = data[:lanes * per].reshape(lanes, per // 8, 8)
        st = np.zeros(lanes, np.uint32)
        st[0] = c                                           # only the first lane continues the running CRC
        for j in range(per // 8):
            w = body[:, j, :].astype(np.uint32)
            lo = st ^ (w[:, 0] | (w[:, 1] << 8) | (w[:, 2] << 16) | (w[:, 3] << 24))
            st = (t[7][lo & 0xFF] ^ t[6][(lo >> 8) & 0xFF] ^ t[5][(lo >> 16) & 0xFF] ^ t[4][lo >> 24] ^
                  t[3][w[:, 4]] ^ t[2][w[:, 5]] ^ t[1][w[:, 6]] ^ t[0][w[:, 7]])
        # stitch: advance lane i's register over the bytes of the lanes behind it = feed `per` zero bytes, lane by lane
        zero_step = _zero_operator(per)
        c = int(st[0])
        for i in range(1, lanes):
            c = _apply_operator(zero_step, c) ^ int(st[i])
    tab = _TABLE_LIST
    for b in data[lanes * per:].tolist():
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _apply_operator(op, c):
    out = 0
    i = 0
    while c:
        if c & 1:
            out ^= op[i]
        c >>= 1
        i += 1
    return out


_ZERO_OPS = {}


def _zero_operator(nbytes):
    """32x32 GF(2) matrix (as 32 column words) that advances a CRC register over `nbytes` zero bytes."""
    if nbytes in _ZERO_OPS:
        return _ZERO_OPS[nbytes]
    one = []                                                # one zero BYTE: register bit i -> table walk of that bit
    for i in range(32):
        c = 1 << i
        c = _TABLE_LIST[c & 0xFF] ^ (c >> 8)
        one.append(c)
    def mul(a, b):                                          # (a o b)(x) = a(b(x))
        return [_apply_operator(a, col) for col in b]
    result = None
    power, n = one, nbytes
    while n:
        if n & 1:
            result = power if result is None else mul(power, result)
        power = mul(power, power)
        n >>= 1
    _ZERO_OPS[nbytes] = result
    return result


_native = False


def _native_crc():
    global _native
    if _native is False:
        _native = None
        try:
            import ctypes as C
            from . import _lib
            l = _lib.lib()
            fn = l.p3d_crc32c
            fn.restype = C.c_uint32
            fn.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]

            def run(mv, crc):
                a = np.frombuffer(mv, dtype=np.uint8)
                return int(fn(a.ctypes.data, a.size, crc))
            _native = run
        except Exception:
            _native = None
    return _native


def mask_crc(crc):
    return (((crc >> 15) | (crc << 17)) + _MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(masked):
    rot = (masked - _MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ---- protobuf wire format (just what the two bundle messages need) ------------------------------------------------------
def _varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    shift = 0
    val = 0
    while True:
        if pos >= len(buf):
            raise ValueError("truncated varint")
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint too long")


def _fields(buf):
    """[(field number, wire type, value)] of one protobuf message; value is an int (varint / fixed) or bytes."""
    pos = 0
    out = []
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]; pos += 8
        elif wt == 2:
            n, pos = _read_varint(buf, pos)
            v = bytes(buf[pos:pos + n]); pos += n
            if len(v) != n:
                raise ValueError("truncated length-delimited field")
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]; pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.append((num, wt, v))
    return out


def _shape_proto(shape):
    # TensorShapeProto { repeated Dim dim = 2 { int64 size = 1 } }
    out = b""
    for d in shape:
        dim = b"\x08" + _varint(int(d))
        out += b"\x12" + _varint(len(dim)) + dim
    return out


def _parse_shape(buf):
    dims = []
    for num, wt, v in _fields(buf):
        if num == 2 and wt == 2:
            size = 0
            for n2, w2, v2 in _fields(v):
                if n2 == 1 and w2 == 0:
                    size = v2 if v2 < (1 << 63) else v2 - (1 << 64)
            dims.append(size)
        elif num == 3 and wt == 0 and v:
            raise ValueError("tensor of unknown rank in checkpoint")
    return tuple(dims)


def _entry_proto(dtype, shape, shard, offset, size, crc_masked):
    # BundleEntryProto { dtype = 1; shape = 2; shard_id = 3; offset = 4; size = 5; fixed32 crc32c = 6; }
    sp = _shape_proto(shape)
    out = b"\x08" + _varint(dtype) + b"\x12" + _varint(len(sp)) + sp
    if shard:
        out += b"\x18" + _varint(shard)
    if offset:
        out += b"\x20" + _varint(offset)
    if size:
        out += b"\x28" + _varint(size)
    out += b"\x35" + struct.pack("<I", crc_masked)
    return out


def _header_proto(num_shards):
    # BundleHeaderProto { num_shards = 1; endianness = 2 (LITTLE = 0, omitted); VersionDef version = 3 { producer = 1 } }
    return b"\x08" + _varint(num_shards) + b"\x1a\x02\x08\x01"


# ---- the table file (<prefix>.index) --------------------------------------------------------------------------------------
def _block(entries, restart_interval=16):
    """One table block: prefix-compressed (key, value) entries + restart array.  `entries` sorted by key."""
    out = bytearray()
    restarts = []
    last = b""
    for i, (k, v) in enumerate(entries):
        if i % restart_interval == 0:
            restarts.append(len(out))
            shared = 0
        else:
            shared = 0
            m = min(len(last), len(k))
            while shared < m and last[shared] == k[shared]:
                shared += 1
        out += _varint(shared) + _varint(len(k) - shared) + _varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _short_successor(key):
    """table_builder.cc Finish(): the index key of the LAST data block is the shortest key >= the block's last key --
    its first byte that is not 0xff incremented, the rest dropped (BytewiseComparator::FindShortSuccessor)."""
    for i, b in enumerate(key):
        if b != 0xFF:
            return key[:i] + bytes([b + 1])
    return key


def _shortest_separator(start, limit):
    """table_builder.cc Add(): the index key of a data block followed by another one is the shortest key in
    [last key of the block, first key of the next) (BytewiseComparator::FindShortestSeparator)."""
    m = min(len(start), len(limit))
    d = 0
    while d < m and start[d] == limit[d]:
        d += 1
    if d < m and start[d] < 0xFF and start[d] + 1 < limit[d]:
        return start[:d] + bytes([start[d] + 1])
    return start


def _with_trailer(block):
    # 1 byte compression type (0 = none) + masked crc32c over block + type byte
    crc = mask_crc(crc32c(block + b"\x00"))
    return block + b"\x00" + struct.pack("<I", crc)


def _parse_block(raw):
    """[(key, value)] of one (uncompressed, trailer-less) block."""
    if len(raw) < 4:
        raise ValueError("table block too short")
    n_restarts = struct.unpack_from("<I", raw, len(raw) - 4)[0]
    end = len(raw) - 4 - 4 * n_restarts
    if end < 0:
        raise ValueError("corrupt restart array")
    pos = 0
    key = b""
    out = []
    while pos < end:
        shared, pos = _read_varint(raw, pos)
        non_shared, pos = _read_varint(raw, pos)
        vlen, pos = _read_varint(raw, pos)
        if shared > len(key) or pos + non_shared + vlen > end:
            raise ValueError("corrupt table entry")
        key = key[:shared] + bytes(raw[pos:pos + non_shared]); pos += non_shared
        out.append((key, bytes(raw[pos:pos + vlen]))); pos += vlen
    return out


def _read_block(buf, offset, size, verify=True):
    raw = buf[offset:offset + size]
    trailer = buf[offset + size:offset + size + 5]
    if len(raw) != size or len(trailer) != 5:
        raise ValueError("table block beyond end of file")
    if trailer[0] == 1:
        raise NotImplementedError("snappy-compressed index blocks are not supported (tf.train.Saver writes them uncompressed)")
    if trailer[0] != 0:
        raise ValueError("unknown block compression type %d" % trailer[0])
    if verify and unmask_crc(struct.unpack_from("<I", trailer, 1)[0]) != crc32c(bytes(raw) + b"\x00"):
        raise ValueError("index block checksum mismatch")
    return _parse_block(raw)


def _read_table(path, verify=True):
    buf = open(path, "rb").read()
    if len(buf) < 48:
        raise ValueError("%s is too short to be a checkpoint index" % path)
    footer = buf[-48:]
    if struct.unpack_from("<Q", footer, 40)[0] != TABLE_MAGIC:
        raise ValueError("%s: bad table magic (not a TensorFlow V2 checkpoint index)" % path)
    pos = 0
    _, pos = _read_varint(footer, pos)      # metaindex handle (unused)
    _, pos = _read_varint(footer, pos)
    idx_off, pos = _read_varint(footer, pos)
    idx_size, pos = _read_varint(footer, pos)
    entries = []
    for _, handle in _read_block(buf, idx_off, idx_size, verify):
        off, p = _read_varint(handle, 0)
        size, p = _read_varint(handle, p)
        entries.extend(_read_block(buf, off, size, verify))
    return entries


# ---- public API ---------------------------------------------------------------------------------------------------------------
def data_file(prefix, shard, num_shards):
    return "%s.data-%05d-of-%05d" % (prefix, shard, num_shards)


def list_variables(prefix, verify=True):
    """[(name, shape, numpy dtype)] in the bundle's (sorted) order -- tf.train.list_variables."""
    out = []
    for key, val in _read_table(prefix + ".index", verify):
        if key == b"":
            continue
        dtype, shape = None, ()
        for num, wt, v in _fields(val):
            if num == 1 and wt == 0:
                dtype = v
            elif num == 2 and wt == 2:
                shape = _parse_shape(v)
        out.append((key.decode(), shape, _DTYPES.get(dtype)))
    return out


def read_checkpoint(prefix, verify=True, names=None):
    """{variable name: numpy array} of a TF-1.x V2 checkpoint `<prefix>.index` + `<prefix>.data-*`.
    verify: check the CRC-32C of every index block and every tensor, as TensorFlow's BundleReader does."""
    table = _read_table(prefix + ".index", verify)
    header = [v for k, v in table if k == b""]
    if not header:
        raise ValueError("%s.index has no bundle header" % prefix)
    num_shards, endian = 1, 0
    for num, wt, v in _fields(header[0]):
        if num == 1 and wt == 0:
            num_shards = v
        elif num == 2 and wt == 0:
            endian = v
    if endian != 0:
        raise NotImplementedError("big-endian checkpoints are not supported")
    shards = {}
    out = {}
    for key, val in table:
        if key == b"":
            continue
        name = key.decode()
        if names is not None and name not in names:
            continue
        dtype = shard = offset = size = 0
        crc = None
        shape = ()
        for num, wt, v in _fields(val):
            if num == 1 and wt == 0: dtype = v
            elif num == 2 and wt == 2: shape = _parse_shape(v)
            elif num == 3 and wt == 0: shard = v
            elif num == 4 and wt == 0: offset = v
            elif num == 5 and wt == 0: size = v
            elif num == 6 and wt == 5: crc = v
            elif num == 7: raise NotImplementedError("variable %s is partitioned (slices): not written by the reference" % name)
        if dtype not in _DTYPES:
            raise NotImplementedError("variable %s has TensorFlow dtype %d; only float32 / int32 / int64 are supported" % (name, dtype))
        dt = _DTYPES[dtype]
        count = int(np.prod(shape)) if shape else 1
        if size != count * dt.itemsize:
            raise ValueError("variable %s: %d bytes stored for shape %s" % (name, size, shape))
        if shard not in shards:
            shards[shard] = np.memmap(data_file(prefix, shard, num_shards), dtype=np.uint8, mode="r")
        raw = shards[shard][offset:offset + size]
        if len(raw) != size:
            raise ValueError("variable %s lies beyond the end of its data file" % name)
        if verify and crc is not None and unmask_crc(crc) != crc32c(raw):
            raise ValueError("variable %s: checksum mismatch" % name)
        out[name] = np.frombuffer(bytes(raw), dtype=dt).reshape(shape).astype(dt.newbyteorder("="))
    return out


def write_checkpoint(prefix, variables, block_bytes=262144):
    """Write {name: array} as `<prefix>.index` + `<prefix>.data-00000-of-00001`, the layout tf.train.Saver(write_version=V2)
    produces for unpartitioned variables (entries sorted by name, tensors back to back in that order).  The table follows
    table_builder.cc step by step: a data block is closed once its size estimate (entries + restart array + count) reaches
    table::Options::block_size (262144 in TensorFlow), its index key is the shortest separator to the next block's first
    key (the short successor of its last key for the final block), restart intervals 16 / 1, no compression."""
    names = sorted(variables)
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    entries = [(b"", _header_proto(1))]
    offset = 0
    with open(data_file(prefix, 0, 1), "wb") as f:
        for n in names:
            a = np.asarray(variables[n], order="C")          # (ascontiguousarray would turn a scalar into shape (1,))
            if a.dtype not in _DT_OF:
                raise TypeError("variable %s has dtype %s; float32 / int32 / int64 only" % (n, a.dtype))
            raw = a.astype(a.dtype.newbyteorder("<")).tobytes()
            f.write(raw)
            entries.append((n.encode(), _entry_proto(_DT_OF[a.dtype], a.shape, 0, offset, len(raw), mask_crc(crc32c(raw)))))
            offset += len(raw)
    # data blocks of ~block_bytes, then metaindex (empty), index, footer
    out = bytearray()
    index = []
    cur = []
    pending = None                      # (last key, handle) of a closed block whose index entry waits for the next key

    def flush():
        nonlocal cur, pending
        blk = _block(cur)
        pending = (cur[-1][0], _varint(len(out)) + _varint(len(blk)))
        out.extend(_with_trailer(blk))
        cur = []

    # BlockBuilder::CurrentSizeEstimate() = entry bytes + 4 per restart point + 4 (the restart count), kept as a running sum
    # (re-encoding the open block per entry made the index build quadratic in the number of variables)
    est, last = 8, b""
    for k, v in entries:
        if pending:
            index.append((_shortest_separator(pending[0], k), pending[1]))
            pending = None
        shared = 0
        if len(cur) % 16 == 0:
            if cur:
                est += 4                   # a new restart point (the first one is in the 8 above)
        else:
            m = min(len(last), len(k))
            while shared < m and last[shared] == k[shared]:
                shared += 1
        est += len(_varint(shared)) + len(_varint(len(k) - shared)) + len(_varint(len(v))) + len(k) - shared + len(v)
        cur.append((k, v))
        last = k
        if est >= block_bytes:
            flush()
            est, last = 8, b""
    if cur:
        flush()
    if pending:
        index.append((_short_successor(pending[0]), pending[1]))
    meta = _block([])
    meta_handle = _varint(len(out)) + _varint(len(meta))
    out.extend(_with_trailer(meta))
    idx = _block(index, restart_interval=1)
    idx_handle = _varint(len(out)) + _varint(len(idx))
    out.extend(_with_trailer(idx))
    footer = meta_handle + idx_handle
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    out.extend(footer)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))


def update_checkpoint_state(directory, prefix, keep=10):
    """The `checkpoint` text file tf.train.Saver maintains next to the bundles (CheckpointState in text format):
    newest prefix first, at most `keep` (train.py:185 max_to_keep=10) -- read back by get_checkpoint_state."""
    path = os.path.join(directory, "checkpoint")
    old = all_checkpoints(directory)
    rel = os.path.basename(prefix)
    paths = [p for p in old if p != rel] + [rel]
    dropped, paths = paths[:-keep], paths[-keep:]
    with open(path, "w") as f:
        f.write('model_checkpoint_path: "%s"\n' % rel)
        for p in paths:
            f.write('all_model_checkpoint_paths: "%s"\n' % p)
    import glob
    for p in dropped:                                   # Saver deletes what falls out of the window: exactly the files of that prefix
        base = os.path.join(directory, p)               # (not `<prefix>.*`: `model.best.index` or a `<prefix>.metrics.json` are not ours)
        for fn in [base + ".index", base + ".meta"] + glob.glob(glob.escape(base) + ".data-[0-9][0-9][0-9][0-9][0-9]-of-[0-9][0-9][0-9][0-9][0-9]"):
            try:
                os.remove(fn)
            except OSError:
                pass


def all_checkpoints(directory):
    path = os.path.join(directory, "checkpoint")
    out = []
    if os.path.exists(path):
        for line in open(path):
            line = line.strip()
            if line.startswith("all_model_checkpoint_paths:"):
                out.append(line.split(":", 1)[1].strip().strip('"'))
    return out


def latest_checkpoint(directory):
    """tf.train.get_checkpoint_state(dir).model_checkpoint_path (train.py:206-209, gen_pred.py:58-62), or None."""
    path = os.path.join(directory, "checkpoint")
    if not os.path.exists(path):
        return None
    for line in open(path):
        line = line.strip()
        if line.startswith("model_checkpoint_path:"):
            p = line.split(":", 1)[1].strip().strip('"')
            return p if os.path.isabs(p) else os.path.join(directory, p)
    return None
