"""Snappy pages inflated on the device (fdb_snappy_decode_pages, snappy_decode_kernel): bit-identical to pyarrow's codec on payloads
that exercise every element kind — literals with 1 … 4 length bytes, copies with 1-, 2- and 4-byte offsets, overlapping patterns of
every short period, empty and one-byte pages — many pages per launch, and damaged pages refused one by one without touching the others."""
import numpy as np
import pyarrow as pa
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pp():
    from frostdb_amd import build
    build.build()
    from frostdb_amd import physicalplan
    return physicalplan


def _payloads():
    rng = np.random.default_rng(11)
    out = [b"", b"x", b"ab" * 3, bytes(range(60)), bytes(range(61)), bytes(rng.integers(0, 256, 59, dtype=np.uint8)), bytes(rng.integers(0, 256, 300, dtype=np.uint8)),
           bytes(rng.integers(0, 256, 70_000, dtype=np.uint8)),             # incompressible: literals with 2- and 3-byte lengths
           bytes(rng.integers(0, 256, 1 << 20, dtype=np.uint8)),            # 1 MiB of noise (a DOUBLE page of random values looks like this)
           b"\x00" * 100_000, b"\x07" * 17, b"abc" * 50_000, b"0123456" * 9_999, bytes(range(256)) * 300,  # patterns of period 1, 3, 7, 256
           np.arange(200_000, dtype=np.int64).tobytes(),                     # a timestamp-like column: long matches at offset 8 … 64
           (1_700_000_000_000 + 15_000 * (np.arange(131_072) // 7)).astype(np.int64).tobytes(),
           rng.integers(0, 6, 500_000).astype(np.uint32).tobytes(),         # dictionary indices: short matches, short literals
           b" ".join(b"/api/v1/p%04d" % rng.integers(0, 1000) for _ in range(40_000))]
    big = bytearray(rng.integers(0, 256, 300_000, dtype=np.uint8).tobytes())
    big[200_000:260_000] = big[0:60_000]  # a match 200 000 bytes back: 4-byte (or 2-byte, fragment-local) offsets
    out.append(bytes(big))
    return out


def test_device_snappy_equals_the_codec(pp):
    codec = pa.Codec("snappy")
    plain = _payloads()
    comp = [codec.compress(p, asbytes=True) for p in plain]
    out, status, ms = pp.snappy_decode_pages(comp, [len(p) for p in plain])
    assert status == [0] * len(plain), status
    for i, (a, b) in enumerate(zip(out, plain)):
        assert a == b, (i, len(b))
    # every page alone, too (page offsets and the window refill at other alignments)
    for i, (c, p) in enumerate(zip(comp, plain)):
        o, st, _ = pp.snappy_decode_pages([c], [len(p)])
        assert st == [0] and o[0] == p, i


def test_device_snappy_hand_made_elements(pp):
    """Streams no compressor emits but the format allows: a copy with a 4-byte offset, a literal with a 4-byte length, a pattern copy
    longer than 64 bytes with offset 1 … 9, a copy whose source ends exactly where the destination starts."""
    def varint(n):
        b = bytearray()
        while True:
            b.append((n & 0x7F) | (0x80 if n > 0x7F else 0))
            n >>= 7
            if not n:
                return bytes(b)

    def lit(data, nbytes=None):
        l = len(data) - 1
        if nbytes is None and l < 60:
            return bytes([l << 2]) + data
        nb = nbytes or (1 if l < 256 else 2 if l < 65536 else 3)
        return bytes([(59 + nb) << 2]) + l.to_bytes(nb, "little") + data

    def copy4(length, off):
        return bytes([((length - 1) << 2) | 3]) + off.to_bytes(4, "little")

    def copy2(length, off):
        return bytes([((length - 1) << 2) | 2]) + off.to_bytes(2, "little")

    def copy1(length, off):
        return bytes([((off >> 8) << 5) | ((length - 4) << 2) | 1, off & 0xFF])

    cases = []
    seed = bytes(range(1, 10))
    for off in range(1, 10):
        body = lit(seed) + copy2(64, off) + copy2(64, off) + copy1(11, off) + copy4(33, off)
        want = bytearray(seed)
        for ln in (64, 64, 11, 33):
            for _ in range(ln):
                want.append(want[-off])
        cases.append((varint(len(want)) + body, bytes(want)))
    data = bytes(np.random.default_rng(3).integers(0, 256, 1000, dtype=np.uint8))
    cases.append((varint(2000) + lit(data, nbytes=4) + copy4(64, 1000) + copy2(64, 1000) + copy4(64, 1000) * 13 + copy2(40, 1000), data + data))
    comp, plain = [c for c, _ in cases], [p for _, p in cases]
    out, status, _ = pp.snappy_decode_pages(comp, [len(p) for p in plain])
    assert status == [0] * len(cases), status
    assert out == plain


def test_device_snappy_refuses_damaged_pages_one_by_one(pp):
    codec = pa.Codec("snappy")
    good = np.arange(50_000, dtype=np.int64).tobytes()
    c = codec.compress(good, asbytes=True)
    bad_len = bytes([c[0] ^ 1]) + c[1:]                      # another length in the preamble
    truncated = c[:len(c) // 2]
    bad_offset = c[:1 + (1 if c[0] < 0x80 else 2 if c[1] < 0x80 else 3)] + bytes([0x02 | (10 << 2), 0xFF, 0xFF]) + c[8:]  # a copy from before the page's first byte
    pages = [c, bad_len, c, truncated, bad_offset, c]
    out, status, _ = pp.snappy_decode_pages(pages, [len(good)] * len(pages))
    assert status[0] == status[2] == status[5] == 0 and out[0] == out[2] == out[5] == good
    assert status[1] == 1 and status[3] in (2, 5) and status[4] != 0, status
    assert out[1] is None and out[3] is None and out[4] is None


def test_device_snappy_rate(pp, capsys):
    """What a launch over a row group's pages reaches (reported, loosely bounded): 240 pages of 1 MiB — a third noise (DOUBLE values),
    a third a DELTA-friendly int64 column, a third dictionary indices."""
    codec = pa.Codec("snappy")
    rng = np.random.default_rng(1)
    plain = []
    for k in range(240):
        if k % 3 == 0:
            plain.append(rng.uniform(0, 1000, 131_072).tobytes())
        elif k % 3 == 1:
            plain.append((1_700_000_000_000 + 15_000 * (np.arange(131_072) // 7 + k)).astype(np.int64).tobytes())
        else:
            plain.append(rng.integers(0, 6, 262_144).astype(np.uint32).tobytes())
    comp = [codec.compress(p, asbytes=True) for p in plain]
    out, status, ms = pp.snappy_decode_pages(comp, [len(p) for p in plain])
    assert status == [0] * 240 and out == plain
    out, status, ms = pp.snappy_decode_pages(comp, [len(p) for p in plain])
    gb = sum(len(p) for p in plain) / 1e9
    with capsys.disabled():
        print(f"\n[snappy] 240 pages, {sum(len(c) for c in comp) / 1e6:.0f} MB -> {gb * 1e3:.0f} MB in {ms:.3f} ms = {gb / (ms * 1e-3):.1f} GB/s of output")
        for kind, name in enumerate(("noise (float64 values)", "int64 timestamps", "dictionary indices")):
            cs, ps = comp[kind::3], plain[kind::3]
            pp.snappy_decode_pages(cs, [len(p) for p in ps])
            _, st, ms_k = pp.snappy_decode_pages(cs, [len(p) for p in ps])
            assert st == [0] * len(cs)
            print(f"[snappy]   80 pages of {name}: {sum(len(c) for c in cs) / 1e6:.1f} MB -> {sum(len(p) for p in ps) / 1e6:.0f} MB in {ms_k:.3f} ms"
                  f" = {sum(len(p) for p in ps) / 1e9 / (ms_k * 1e-3):.1f} GB/s of output, {sum(len(p) for p in ps) / len(ps) / 1e6 / (ms_k * 1e-3):.0f} MB/s per page")
    assert ms < 200.0
