"""the TF-1.x checkpoint bundle the reference's Saver writes (train.py:180-185,204-210,
266-267).  No TensorFlow-written file exists here (parity unpinned, see sap3d_tensorflow_amd/tf_checkpoint.py); what IS
pinned: CRC-32C known answers, the published constants of the table format, the protobuf wire layout of a hand-assembled
entry, round trips, and failure on corrupted files."""
import os
import struct

import numpy as np
import pytest

from sap3d_tensorflow_amd import tf_checkpoint as tfc


def test_crc32c_known_answers():
    # RFC 3720 appendix B.4 test vectors for CRC-32C
    assert tfc.crc32c(b"123456789") == 0xE3069283
    assert tfc.crc32c(bytes(32)) == 0x8A9136AA
    assert tfc.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert tfc.crc32c(bytes(range(32))) == 0x46DD794E
    # running value and masking (leveldb: rotate right 15, add 0xa282ead8)
    assert tfc.crc32c(b"6789", tfc.crc32c(b"12345")) == 0xE3069283
    assert tfc.mask_crc(0xE3069283) == (((0xE3069283 >> 15) | (0xE3069283 << 17)) + 0xA282EAD8) & 0xFFFFFFFF
    assert tfc.unmask_crc(tfc.mask_crc(0x12345678)) == 0x12345678
    big = np.random.default_rng(0).integers(0, 256, 300_000, dtype=np.uint8).tobytes()      # takes the native path when built
    c = 0
    for i in range(0, len(big), 4096):
        c = tfc.crc32c(big[i:i + 4096], c)
    assert tfc.crc32c(big) == c


def test_wire_layout_of_an_entry():
    # BundleEntryProto{dtype: DT_FLOAT, shape {dim{size:3} dim{size:5}}, offset: 300, size: 60, crc32c: 0x01020304}
    got = tfc._entry_proto(1, (3, 5), 0, 300, 60, 0x01020304)
    want = bytes([0x08, 0x01, 0x12, 0x08, 0x12, 0x02, 0x08, 0x03, 0x12, 0x02, 0x08, 0x05, 0x20, 0xAC, 0x02, 0x28, 0x3C, 0x35,
                  0x04, 0x03, 0x02, 0x01])
    assert got == want
    assert tfc._header_proto(1) == bytes([0x08, 0x01, 0x1A, 0x02, 0x08, 0x01])


def _model(rng):
    return {
        "firstconv1": rng.standard_normal((1, 7, 7, 3, 64)).astype(np.float32),
        "batch_normalization/gamma": rng.standard_normal(64).astype(np.float32),
        "batch_normalization/moving_variance": rng.random(64).astype(np.float32),
        "conv3_0_1": rng.standard_normal((1, 1, 1, 64, 64)).astype(np.float32),
        "conv3d_transpose_2/bias": np.float32(0.25).reshape(()),          # rank 0
        "global_step": np.int64(1234).reshape(()),
        **{"block%d/w" % i: rng.standard_normal((3, i + 1)).astype(np.float32) for i in range(300)},      # several index blocks
    }


def test_round_trip_and_state_file(tmp_path):
    rng = np.random.default_rng(1)
    m = _model(rng)
    prefix = str(tmp_path / "model" / "run" / "p3d_1000.ckpt")
    tfc.write_checkpoint(prefix, m, block_bytes=4096)      # several data blocks: separator keys in the index block
    assert os.path.exists(prefix + ".index") and os.path.exists(prefix + ".data-00000-of-00001")
    raw = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == 0xDB4775248B80FB57
    assert os.path.getsize(prefix + ".data-00000-of-00001") == sum(v.nbytes for v in m.values())
    back = tfc.read_checkpoint(prefix)
    assert sorted(back) == sorted(m)
    for k, v in m.items():
        assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v), k
    listed = tfc.list_variables(prefix)
    assert [n for n, _, _ in listed] == sorted(m)
    assert dict((n, s) for n, s, _ in listed)["firstconv1"] == (1, 7, 7, 3, 64)
    only = tfc.read_checkpoint(prefix, names={"conv3_0_1"})
    assert list(only) == ["conv3_0_1"]
    # Saver bookkeeping: newest first, max_to_keep
    d = os.path.dirname(prefix)
    for step in (1000, 2000, 3000):
        p = os.path.join(d, "p3d_%d.ckpt" % step)
        tfc.write_checkpoint(p, {"v": np.float32([step])})
        tfc.update_checkpoint_state(d, p, keep=2)
    assert tfc.latest_checkpoint(d) == os.path.join(d, "p3d_3000.ckpt")
    assert tfc.all_checkpoints(d) == ["p3d_2000.ckpt", "p3d_3000.ckpt"]
    assert not os.path.exists(os.path.join(d, "p3d_1000.ckpt.index"))
    assert tfc.latest_checkpoint(str(tmp_path)) is None


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / "c.ckpt")
    tfc.write_checkpoint(prefix, {"a": np.arange(100, dtype=np.float32), "b": np.ones((4, 4), np.float32)})
    data = prefix + ".data-00000-of-00001"
    raw = bytearray(open(data, "rb").read())
    raw[17] ^= 0x40
    open(data, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        tfc.read_checkpoint(prefix)
    assert tfc.read_checkpoint(prefix, verify=False)["b"].shape == (4, 4)
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[3] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(ValueError):
        tfc.read_checkpoint(prefix)
    open(prefix + ".index", "wb").write(b"not a table")
    with pytest.raises(ValueError):
        tfc.read_checkpoint(prefix)


@pytest.mark.gpu
def test_session_restores_a_bundle(tmp_path):
    """P3DSession.save_checkpoint / restore: a bundle holding every variable of the graph under its TF name round-trips
    through the HIP session, from a directory (checkpoint state file) and from a prefix."""
    from oracle import p3d
    from sap3d_tensorflow_amd import P3DSession
    cfg = p3d.NetConfig(base=16, blocks=(1, 1, 2))
    shape = (2, 16, 32, 32)
    a = P3DSession("unet", batch=2, frames=16, height=32, width=32, base=cfg.base, blocks=cfg.blocks, seed=3)
    d = str(tmp_path / "model" / "run")
    prefix = a.save_checkpoint(d, 7)
    names = [n for n, _, _ in a.variables()]
    assert [n for n, _, _ in tfc.list_variables(prefix)] == sorted(names)
    x = p3d.synthetic_clip(0, shape + (3,))
    want = a.forward(x, 0.0, False)
    b = P3DSession("unet", batch=2, frames=16, height=32, width=32, base=cfg.base, blocks=cfg.blocks, seed=99)
    assert not np.array_equal(b.forward(x, 0.0, False), want)
    assert b.restore(d) == prefix
    assert np.array_equal(b.forward(x, 0.0, False), want)
    c = P3DSession("concat", batch=2, frames=16, height=32, width=32, base=cfg.base, blocks=cfg.blocks, seed=1)
    with pytest.raises(KeyError):
        c.restore(prefix)            # another head: variables missing from the bundle
    a.close(); b.close(); c.close()


def test_numpy_crc_fallback_matches_the_byte_loop():
    """The fallback used when libp3dhip cannot be loaded (256 lanes, slicing-by-8, stitched with the CRC's zero-extension
    operator) against the byte-at-a-time definition, with and without a running value, ragged tails included."""
    rng = np.random.default_rng(0)

    def slow(b, crc=0):
        c = crc ^ 0xFFFFFFFF
        for x in b:
            c = tfc._TABLE_LIST[(c ^ x) & 0xFF] ^ (c >> 8)
        return c ^ 0xFFFFFFFF

    for n in (4096, 4097, 5000, 70001):
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert tfc._crc32c_numpy(memoryview(b), 0) == slow(b)
        assert tfc._crc32c_numpy(memoryview(b), 0xDEADBEEF) == slow(b, 0xDEADBEEF)
    assert tfc._crc32c_numpy(memoryview(b"123456789" * 1000), 0) == slow(b"123456789" * 1000)


def test_dropped_checkpoints_lose_every_file_of_their_prefix(tmp_path):
    """tf.train.Saver(max_to_keep) removes whatever belongs to a dropped prefix: .index, every data shard, a TF-written .meta."""
    d = str(tmp_path)
    for step in (1, 2, 3):
        prefix = os.path.join(d, "p3d_%d.ckpt" % step)
        tfc.write_checkpoint(prefix, {"w": np.arange(4, dtype=np.float32)})
        open(prefix + ".meta", "w").write("graph")                 # what a TensorFlow-written checkpoint also leaves
        open(prefix + ".data-00001-of-00002", "w").write("shard")
        tfc.update_checkpoint_state(d, prefix, keep=2)
        open(prefix + ".metrics.json", "w").write("{}")            # a sibling that merely starts with the prefix: not Saver's to delete
    left = sorted(os.listdir(d))
    assert [f for f in left if f.startswith("p3d_1.ckpt")] == ["p3d_1.ckpt.metrics.json"], left
    assert [f for f in left if f.startswith("p3d_2.ckpt")] and [f for f in left if f.startswith("p3d_3.ckpt")]


def _crc32c_bitwise(data):
    """CRC-32C straight from its definition (reflected polynomial 0x82F63B78, one bit at a time): shares nothing with the
    table-driven code under test."""
    crc = 0xFFFFFFFF
    for b in data:
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


def _masked(data):
    c = _crc32c_bitwise(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def test_index_file_is_byte_for_byte_what_the_format_prescribes(tmp_path):
    """A two-variable bundle assembled BY HAND from the published format (tensor_bundle.proto, table_builder.cc, format.cc,
    crc32c.h) -- not from a TensorFlow-written file, none exists here -- against what write_checkpoint emits.  Guards the
    writer against a bug that its own reader would mirror."""
    import struct
    a = np.array([1.0, 2.0], np.float32)
    c = np.int32(7).reshape(())
    prefix = str(tmp_path / "g.ckpt")
    tfc.write_checkpoint(prefix, {"a": a, "b/c": c})
    a_raw = struct.pack("<2f", 1.0, 2.0)
    c_raw = struct.pack("<i", 7)
    assert open(prefix + ".data-00000-of-00001", "rb").read() == a_raw + c_raw
    # values: BundleHeaderProto{num_shards: 1, version{producer: 1}}; BundleEntryProto per tensor (zero-valued fields omitted)
    header = bytes([0x08, 0x01, 0x1A, 0x02, 0x08, 0x01])
    ent_a = bytes([0x08, 0x01,                            # dtype DT_FLOAT
                   0x12, 0x04, 0x12, 0x02, 0x08, 0x02,     # shape { dim { size: 2 } }
                   0x28, 0x08,                             # size 8 (offset 0 omitted)
                   0x35]) + struct.pack("<I", _masked(a_raw))
    ent_c = bytes([0x08, 0x03,                            # dtype DT_INT32
                   0x12, 0x00,                             # shape {} (rank 0)
                   0x20, 0x08,                             # offset 8
                   0x28, 0x04,                             # size 4
                   0x35]) + struct.pack("<I", _masked(c_raw))
    # data block: entries (shared, non-shared, value length, key suffix, value), one restart at 0, restart count
    data = bytes([0, 0, len(header)]) + header
    data += bytes([0, 1, len(ent_a)]) + b"a" + ent_a
    data += bytes([0, 3, len(ent_c)]) + b"b/c" + ent_c
    data += struct.pack("<II", 0, 1)
    trailer = lambda blk: blk + b"\x00" + struct.pack("<I", _masked(blk + b"\x00"))      # type 0 = uncompressed, masked CRC
    out = trailer(data)
    meta = struct.pack("<II", 0, 1)                          # empty metaindex block
    meta_off = len(out)
    out += trailer(meta)
    # index block: last data block -> key FindShortSuccessor("b/c") = "c", value = BlockHandle(offset 0, size)
    assert len(data) < 128 and meta_off < 128                # one-byte varints below
    handle = bytes([0, len(data)])
    index = bytes([0, 1, len(handle)]) + b"c" + handle + struct.pack("<II", 0, 1)
    index_off = len(out)
    out += trailer(index)
    footer = bytes([meta_off, len(meta), index_off, len(index)])
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    out += footer
    assert open(prefix + ".index", "rb").read() == out
    got = tfc.read_checkpoint(prefix)
    assert np.array_equal(got["a"], a) and got["b/c"].shape == () and int(got["b/c"]) == 7


def test_index_keys_are_shortest_separators():
    assert tfc._short_successor(b"b/c") == b"c"
    assert tfc._short_successor(b"\xff\xffa") == b"\xff\xffb"
    assert tfc._shortest_separator(b"block12/w", b"block2/w") == b"block12/w"      # '1' + 1 == '2': no room
    assert tfc._shortest_separator(b"abcdef", b"abzz") == b"abd"
    assert tfc._shortest_separator(b"ab", b"abc") == b"ab"                          # a prefix of the limit stays


def test_many_small_variables_write_in_linear_time_and_split_blocks_like_the_full_encoding(tmp_path):
    """the open data block was re-encoded for every entry (10.8 s for 3000 variables).  The running size
    estimate must close blocks exactly where the full encoding would (same index file as a small-block run re-derived with
    _block) and 3000 variables must take well under a second of index building."""
    rng = np.random.default_rng(0)
    vs = {"scope_%04d/layer/kernel" % i: rng.standard_normal(4).astype(np.float32) for i in range(3000)}
    # linear time, stated as WORK rather than as wall clock (a loaded CI box must not fail this: ADVICE round 4): the block
    # encoder runs once per closed data block + the meta-index and index blocks, with every entry encoded exactly once --
    # the quadratic form called it once per entry on the whole open block (3000 calls over ~4.5 million entries)
    calls, encoded = [0], [0]
    real_block = tfc._block

    def counting_block(entries, *a, **k):
        calls[0] += 1
        encoded[0] += len(entries)
        return real_block(entries, *a, **k)

    tfc._block = counting_block
    try:
        tfc.write_checkpoint(str(tmp_path / "big.ckpt"), vs)
    finally:
        tfc._block = real_block
    assert calls[0] <= 8 and encoded[0] <= len(vs) + 16, (calls[0], encoded[0])
    got = tfc.read_checkpoint(str(tmp_path / "big.ckpt"))
    assert sorted(got) == sorted(vs) and all(np.array_equal(got[n], vs[n]) for n in vs)
    # block boundaries: with a small block size, re-derive them with the full encoder and compare the data-block sizes
    tfc.write_checkpoint(str(tmp_path / "small.ckpt"), vs, block_bytes=4096)
    raw = open(str(tmp_path / "small.ckpt.index"), "rb").read()
    entries = [(b"", tfc._header_proto(1))]
    off = 0
    for n in sorted(vs):
        b = vs[n].tobytes()
        entries.append((n.encode(), tfc._entry_proto(tfc._DT_OF[vs[n].dtype], vs[n].shape, 0, off, len(b), tfc.mask_crc(tfc.crc32c(b)))))
        off += len(b)
    want, cur = bytearray(), []
    for e in entries:
        cur.append(e)
        if len(tfc._block(cur)) >= 4096:
            want += tfc._with_trailer(tfc._block(cur)); cur = []
    if cur:
        want += tfc._with_trailer(tfc._block(cur))
    assert raw[:len(want)] == bytes(want)
