"""The one-shot inflater of the PNG path (lilliput_amd/csrc/lp_inflate.cpp) against zlib -- the arbiter it defers to. Its contract:
answer 1 ONLY for a stream zlib accepts, that inflates to exactly the expected number of bytes and ends with its last byte, and then
hand over the same bytes; answer 0 (= "ask zlib") for anything else. No GPU involved."""
import ctypes as C
import zlib

import numpy as np
import pytest
from conftest import fresh_seed

import png_cases


def _fn(L):
    f = L.lilliput_hip_inflate_exact
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
    return f


def _run(f, stream, out_len):
    out = np.zeros(max(out_len, 1), np.uint8)
    r = f(stream, len(stream), out.ctypes.data_as(C.c_void_p), out_len)
    assert r in (0, 1), "the decoder wrote past its output buffer" if r == -1 else r
    return r, out[:out_len].tobytes()


def _payloads():
    rng = np.random.default_rng(5)
    ramp = (np.arange(200000) % 251).astype(np.uint8).tobytes()
    noise = rng.integers(0, 256, 150000, dtype=np.uint8).tobytes()
    photo = np.clip(np.cumsum(rng.integers(-3, 4, 300000)) % 256, 0, 255).astype(np.uint8).tobytes()  # literal-heavy, a few matches
    sparse = bytes(100000) + b"\x01" + bytes(70000)
    text = (b"the quick brown fox jumps over the lazy dog. " * 4000)
    tiny = [b"", b"a", b"ab", b"abc" * 3]
    return {"ramp": ramp, "noise": noise, "photo": photo, "sparse": sparse, "text": text, **{"tiny%d" % i: t for i, t in enumerate(tiny)}}


def test_ordinary_streams_take_the_fast_path_and_match(hip_lib):
    f = _fn(hip_lib)
    for name, data in _payloads().items():
        for level in (0, 1, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED):
                for wbits in (9, 12, 15):
                    co = zlib.compressobj(level, zlib.DEFLATED, wbits, 8, strategy)
                    s = co.compress(data) + co.flush()
                    r, out = _run(f, s, len(data))
                    # Z_HUFFMAN_ONLY / a block without matches can carry a lone distance code (an incomplete set zlib tolerates): that is a 0
                    if r == 1:
                        assert out == data, (name, level, strategy, wbits)
                    else:
                        assert zlib.decompress(s) == data
                        assert strategy == zlib.Z_HUFFMAN_ONLY or len(data) < 16 or name in ("noise",), (name, level, strategy, wbits)


def test_multi_block_and_flush_points(hip_lib):
    f = _fn(hip_lib)
    data = _payloads()["photo"]
    co = zlib.compressobj(6)
    s = b""
    for i in range(0, len(data), 7001):
        s += co.compress(data[i:i + 7001])
        s += co.flush(zlib.Z_SYNC_FLUSH if (i // 7001) % 2 else zlib.Z_FULL_FLUSH)  # empty stored blocks between the others
    s += co.flush()
    r, out = _run(f, s, len(data))
    assert r == 1 and out == data


def test_wrong_size_trailing_bytes_and_bad_checksum_are_not_accepted(hip_lib):
    f = _fn(hip_lib)
    data = _payloads()["text"]
    s = zlib.compress(data, 6)
    assert _run(f, s, len(data))[0] == 1
    assert _run(f, s, len(data) - 1)[0] == 0
    assert _run(f, s, len(data) + 1)[0] == 0
    assert _run(f, s + b"\x00", len(data))[0] == 0
    assert _run(f, s[:-1], len(data))[0] == 0
    bad = bytearray(s); bad[-1] ^= 1
    assert _run(f, bytes(bad), len(data))[0] == 0
    d = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_DEFAULT_STRATEGY, b"dictionary")  # FDICT set
    assert _run(f, d.compress(data) + d.flush(), len(data))[0] == 0


def test_mutated_streams_never_accept_what_zlib_rejects(hip_lib):
    f = _fn(hip_lib)
    rng = np.random.default_rng(fresh_seed(11))
    base = {k: zlib.compress(v, 6) for k, v in _payloads().items() if len(v) > 1000}
    base["fixed"] = (lambda co: co.compress(_payloads()["text"]) + co.flush())(zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED))
    accepted = 0
    for name, s in base.items():
        n = len(zlib.decompress(s))
        for trial in range(400):
            m = bytearray(s)
            kind = trial % 4
            pos = int(rng.integers(0, min(len(m), 400) if kind < 2 else len(m)))  # the block headers and code tables sit at the front
            if kind in (0, 2):
                m[pos] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                m[pos] = int(rng.integers(0, 256))
            else:
                del m[pos:pos + int(rng.integers(1, 4))]
            m = bytes(m)
            r, out = _run(f, m, n)
            if r == 1:
                accepted += 1
                try:
                    ref = zlib.decompress(m)
                except zlib.error as e:
                    raise AssertionError("accepted a stream zlib rejects: %s trial %d: %s" % (name, trial, e))
                assert ref == out, (name, trial)
    assert accepted < 50  # almost every mutation breaks the checksum at least


def _png_filtered(L, data, own):
    L.lilliput_hip_png_set_inflater.restype = C.c_int
    L.lilliput_hip_png_inflate_bytes.restype = C.c_long
    L.lilliput_hip_png_inflate_bytes.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
    prev = L.lilliput_hip_png_set_inflater(1 if own else 0)
    try:
        cap = 1 << 26
        buf = np.zeros(cap, np.uint8)
        r = L.lilliput_hip_png_inflate_bytes(data, len(data), buf.ctypes.data_as(C.c_void_p), cap)
        return None if r < 0 else buf[:r].tobytes()
    finally:
        L.lilliput_hip_png_set_inflater(prev)


def test_png_files_inflate_the_same_through_both_routes(hip_lib):
    """Every PNG fixture and 1 500 damaged variants: the filtered rows (or the rejection) with the library's inflater in front equal
    the answer of the zlib-only route -- which is the one pinned against the reference's libpng (tests/test_png.py)."""
    L = hip_lib
    cases = dict(png_cases.fixtures())
    cases.update(png_cases.generated())
    cases.update(png_cases.fuzz(77, 1500))
    assert len(cases) >= 1500
    differ = [name for name, data in cases.items() if _png_filtered(L, data, True) != _png_filtered(L, data, False)]
    assert not differ, differ[:10]


def test_checksums_equal_zlibs(hip_lib):
    f = hip_lib.lilliput_hip_checksum
    f.restype = C.c_uint32
    f.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_size_t]
    rng = np.random.default_rng(3)
    big = rng.integers(0, 256, 3_000_000 + 77, dtype=np.uint8)
    big[100000:200000] = 255  # the worst case for the Adler sums
    lengths = list(range(0, 200)) + [255, 256, 1023, 4096, 5535, 5536, 5537, 5552, 11072, 65521, 100001, 1 << 20, big.size - 7]
    for n in lengths:
        for off in (0, 1, 7):
            view = big[off:off + n]
            raw = view.tobytes()
            for seed_a, seed_c in ((1, 0), (0x12345678 % 65521 | (4242 << 16), 0xdeadbeef)):
                assert f(0, seed_a, view.ctypes.data_as(C.c_void_p), n) == zlib.adler32(raw, seed_a), ("adler", n, off)
                assert f(1, seed_c, view.ctypes.data_as(C.c_void_p), n) == zlib.crc32(raw, seed_c), ("crc", n, off)
