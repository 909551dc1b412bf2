"""GPU tests of the ingest half of the batched transform (lilliput_hip_batch_transform / lilliput_hip_node_transform): however the
caller's encoded bytes reach the device -- read in place from a pinned arena, from pages registered for the call, or copied through
the engines' pinned slots -- the thumbnails are the reference CPU path's bytes (opencv.cpp:99-171 starts from the caller's []byte;
opencv.go:872-900 is what Encode returns). Also: the headline configuration (default engines, default chunks, 4096 x 4096 sources)
with EVERY output checked, the engine pool of the one-image ABI, and untrusted WebP canvas sizes."""
import ctypes as C
import hashlib
import os
import struct
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _synth_jpegs(seeds, size, quality=90):
    import multiprocessing as mp

    from lilliput_amd import synth

    workers = max(1, min(len(seeds), (os.cpu_count() or 2) - 1, 64))
    with mp.get_context("fork").Pool(workers) as pool:
        return list(pool.imap(synth._job, [(s, size, quality) for s in seeds], chunksize=1))


def _expect(oracle, datas, w, h, q=85):
    from concurrent.futures import ThreadPoolExecutor

    use_ref = oracle.ref() is not None
    with ThreadPoolExecutor(min(64, os.cpu_count() or 4)) as ex:  # ctypes releases the GIL
        return list(ex.map(lambda d: oracle.transform_jpeg_thumbnail(bytes(d), w, h, q, use_ref=use_ref), datas))


@pytest.fixture(scope="module")
def small_set(oracle):
    datas = _synth_jpegs(range(100, 140), 512)
    return datas, _expect(oracle, datas, 64, 64)


def test_every_ingest_route_yields_the_reference_bytes(hip_lib, small_set):
    import lilliput_amd as la

    datas, exp = small_set
    b = la.Batch(0)
    prev = hip_lib.lilliput_hip_set_ingest_mode(b"register")
    try:
        # (1) opt-in: pageable sources registered for the call (512 x 512 q90 sources are ~100 KB: above the 64 KiB registration floor)
        r = b.transform(datas, 64, 64, quality=85)
        st = b.ingest_stats()
        assert [x.status for x in r] == [0] * len(datas)
        assert [x.data for x in r] == exp
        assert st["direct_bytes"] > 0 and st["direct_bytes"] + st["copied_bytes"] == st["staged_bytes"], st
        # (2) everything through the pinned slots
        hip_lib.lilliput_hip_set_ingest_mode(b"staged")
        r = b.transform(datas, 64, 64, quality=85)
        st = b.ingest_stats()
        assert [x.data for x in r] == exp
        assert st["direct_bytes"] == 0 and st["copied_bytes"] == st["staged_bytes"] > 0, st
        # (3) the default: sources in a pinned arena are read in place, nothing registered, nothing copied by the host
        hip_lib.lilliput_hip_set_ingest_mode(b"auto")
        arena = la.HostArena(sum(len(d) + 64 for d in datas) + 4096, 0)
        views = [arena.put(d) for d in datas]
        assert hip_lib.lilliput_hip_host_is_pinned(views[3].ctypes.data, views[3].size) == 1
        r = b.transform(views, 64, 64, quality=85)
        st = b.ingest_stats()
        assert [x.data for x in r] == exp
        assert st["copied_bytes"] == 0 and st["direct_bytes"] == st["staged_bytes"] and st["register_ms"] == 0.0, st
        # sources packed back to back at odd offsets: neighbours in the arena travel as ONE transfer and keep their host spacing on the
        # device, so their entropy-coded segments start at any byte offset (LpJpeg::raw_skip) -- alone, in pairs, in small and odd chunks
        arena2 = la.HostArena(sum(len(d) + 8 for d in datas) + 4096, 0)
        views2 = [arena2.put(d, align=1 + (i % 3)) for i, d in enumerate(datas)]
        for chunk in (0, 1, 2, 7):
            r = b.transform(views2, 64, 64, quality=85, chunk=chunk)
            assert [x.data for x in r] == exp, chunk
        r = b.transform(views2[::-1] + [datas[0]] + views2[5:9], 64, 64, quality=85)  # reversed (no run), a pageable one in between, a short run
        assert [x.data for x in r] == exp[::-1] + [exp[0]] + exp[5:9]
        arena2.close()
        # pageable sources in the default mode are staged, not registered
        r = b.transform(datas[:8], 64, 64, quality=85)
        st = b.ingest_stats()
        assert [x.data for x in r] == exp[:8] and st["direct_bytes"] == 0, st
        arena.close()
    finally:
        hip_lib.lilliput_hip_set_ingest_mode([b"register", b"staged", b"auto"][prev])
        b.close()


def test_duplicate_overlapping_and_tiny_sources_in_one_batch(hip_lib, small_set, fixture_bytes):
    """The same buffer named by several items (round 2's registration attempt aborted the process on that), two items that are views
    of one allocation, sources far below the registration floor, and items sharing pages with registered ones."""
    import lilliput_amd as la

    datas, exp = small_set
    prev = hip_lib.lilliput_hip_set_ingest_mode(b"register")
    a = np.frombuffer(datas[0], dtype=np.uint8).copy()
    big = np.concatenate([np.frombuffer(datas[1], dtype=np.uint8), np.frombuffer(datas[2], dtype=np.uint8)])  # two files in ONE allocation
    v1, v2 = big[: len(datas[1])], big[len(datas[1]):]
    tiny = np.frombuffer(fixture_bytes["sunrise.jpg"], dtype=np.uint8).copy()  # a few KB
    items = [a, a, v1, v2, tiny, a, v2, tiny] * 5
    want = {id(a): exp[0], id(v1): exp[1], id(v2): exp[2]}
    b = la.Batch(0)
    try:
        for chunk in (0, 3):  # chunk 3: the duplicates land in different chunks, staged by different threads at the same time
            r = b.transform(items, 64, 64, quality=85, chunk=chunk)
            assert all(x.status == 0 for x in r)
            tiny_out = {x.data for x, it in zip(r, items) if it is tiny}
            assert len(tiny_out) == 1
            for x, it in zip(r, items):
                if it is not tiny:
                    assert x.data == want[id(it)]
        # nothing stays registered after the call: an explicit long-lived registration of the same buffer must succeed ...
        assert hip_lib.lilliput_hip_host_register(big.ctypes.data, big.size) == 0
        assert hip_lib.lilliput_hip_host_is_pinned(v2.ctypes.data, v2.size) == 1
        r = b.transform([v1, v2, a], 64, 64, quality=85)
        assert [x.data for x in r] == [exp[1], exp[2], exp[0]]
        assert hip_lib.lilliput_hip_host_unregister(big.ctypes.data) == 0
        assert hip_lib.lilliput_hip_host_is_pinned(v2.ctypes.data, v2.size) == 0
        # ... and a read-only mapping (a bytes object's pages may be) or any other refusal falls back to staging, never fails the item
        r = b.transform([bytes(datas[5])], 64, 64, quality=85)
        assert r[0].data == exp[5]
    finally:
        hip_lib.lilliput_hip_set_ingest_mode([b"register", b"staged", b"auto"][prev])
        b.close()


def test_two_batches_at_once_share_registrations(hip_lib, small_set):
    """Two batch objects (two goroutines of a service) transform the SAME source buffers concurrently: the page-range table counts the
    users of a registration, the second call finds the pages pinned and the last one out unregisters."""
    import lilliput_amd as la

    datas, exp = small_set
    prev = hip_lib.lilliput_hip_set_ingest_mode(b"register")
    arrays = [np.frombuffer(d, dtype=np.uint8).copy() for d in datas]
    out = [None, None]

    def work(k):
        b = la.Batch(0)
        for _ in range(3):
            out[k] = [x.data for x in b.transform(arrays, 64, 64, quality=85, chunk=5)]
        b.close()

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    hip_lib.lilliput_hip_set_ingest_mode([b"register", b"staged", b"auto"][prev])
    assert out[0] == exp and out[1] == exp


def test_headline_configuration_every_output_checked(hip_lib, oracle):
    """BASELINE configs[1] as bench.py runs it -- lilliput_hip_batch_transform with the default four engines and 32-image chunks, 8 192-bit
    subsequences, 4096 x 4096 q90 sources in a pinned arena read in place -- on 64 distinct seeds, EVERY thumbnail compared with the
    reference CPU path's bytes; then the same items (pageable copies: the staged route) through the two-slot node entry point."""
    import lilliput_amd as la

    hip_lib.lilliput_hip_set_ingest_mode(b"auto")
    seeds = list(range(2000, 2064))
    datas = _synth_jpegs(seeds, 4096)
    exp = _expect(oracle, datas, 256, 256)
    arena = la.HostArena(sum(len(d) + 64 for d in datas) + 4096, 0)
    arrays = [arena.put(d) for d in datas]
    # 192 items = 6 chunks of 32 over 4 engines, every source three times (the duplicates exercise the page-range table at full size)
    items = arrays * 3
    b = la.Batch(0)
    try:
        b.prepare(items, dst_cap=256 << 10)
        for _ in range(2):
            failed = b.transform_prepared(256, 256, la.ImageOpsFit, False, 85, 0)
            assert failed == 0
            res = b.results()
            bad = [i for i, r in enumerate(res) if r.status != 0 or r.data != exp[i % len(exp)]]
            assert not bad, bad
        st = b.ingest_stats()
        assert st["direct_bytes"] == st["staged_bytes"] > 0 and st["copied_bytes"] == 0, st
    finally:
        b.close()
    n = la.Node([0, 0])
    try:
        res = n.transform([np.frombuffer(d, dtype=np.uint8) for d in datas] * 3, 256, 256, quality=85, dst_cap=256 << 10)
        bad = [i for i, r in enumerate(res) if r.status != 0 or r.data != exp[i % len(exp)]]
        assert not bad, bad
    finally:
        n.close()
        arena.close()


def test_one_image_abi_engines_are_pooled_not_per_thread(hip_lib, fixture_bytes):
    """README.md:82-85 / SURVEY 8(b): handles are used from whatever OS thread cgo picked. 128 threads each run one
    ImageOps.Transform; afterwards the process holds at most the pool's idle engines (8), not one engine per thread."""
    import lilliput_amd as la

    data = fixture_bytes["large-sunrise.jpg"]
    ops0 = la.ImageOps(2048)
    d0 = la.Decoder(data)
    want = ops0.Transform(d0, la.ImageOptions(".jpeg", 128, 128, la.ImageOpsFit, False, {la.JpegQuality: 85}))
    d0.Close()
    ops0.Close()
    free0, total = C.c_size_t(), C.c_size_t()
    assert hip_lib.lilliput_hip_mem_info(0, C.byref(free0), C.byref(total)) == 0
    stats = (C.c_size_t * 4)()
    hip_lib.lilliput_hip_engine_pool_stats(stats)
    created0 = stats[2]
    outs, errs = [None] * 128, []
    gate = threading.Barrier(128)

    def work(i):
        try:
            ops = la.ImageOps(2048)
            dec = la.Decoder(data)
            gate.wait()
            outs[i] = ops.Transform(dec, la.ImageOptions(".jpeg", 128, 128, la.ImageOpsFit, False, {la.JpegQuality: 85}))
            dec.Close()
            ops.Close()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(i,)) for i in range(128)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs[:3]
    assert all(o == want for o in outs)
    hip_lib.lilliput_hip_engine_pool_stats(stats)
    live, idle, created, trimmed = list(stats)
    assert live == 0 and idle <= 64, list(stats)
    assert created - created0 <= 128
    # a handle used from two different threads one call after the other (what a migrating goroutine does): the second call's engine
    # sees what the first one's left on the device
    ops = la.ImageOps(2048)
    dec = la.Decoder(data)
    res = []
    t = threading.Thread(target=lambda: res.append(ops.Transform(dec, la.ImageOptions(".jpeg", 128, 128, la.ImageOpsFit, False, {la.JpegQuality: 85}))))
    t.start()
    t.join()
    dec.Close()
    dec = la.Decoder(data)
    res.append(ops.Transform(dec, la.ImageOptions(".jpeg", 128, 128, la.ImageOpsFit, False, {la.JpegQuality: 85})))
    dec.Close()
    ops.Close()
    assert res == [want, want]
    # device memory: the idle engines' arenas for a 1300 x 1942 decode are tens of MB each; 128 per-thread engines were ~4 GB
    # (calls that were in flight together shared batch launches through the dispatchers of lp_coalesce.h; an idle dispatcher gives its
    # batch back after LILLIPUT_HIP_COALESCE_IDLE_MS, 1 s by default: what the process holds follows the calls in flight)
    import time

    gs = (C.c_size_t * 4)()
    hip_lib.lilliput_hip_guard_stats(gs)
    if gs[0]:
        return  # guard mode (LILLIPUT_HIP_GUARD): buffers are separate mappings of whole pages, hipMemGetInfo is not a measure of the arenas there
    free1 = C.c_size_t()
    for _ in range(40):
        hip_lib.lilliput_hip_mem_info(0, C.byref(free1), C.byref(total))
        if free0.value - free1.value < (2 << 30):
            break
        time.sleep(0.25)
    assert free0.value - free1.value < (2 << 30), (free0.value, free1.value)


def _vp8x_anim_claiming(w, h):
    """A tiny animated WebP whose VP8X header claims a w x h canvas (one 1 x 1 lossless frame)."""
    vp8l = bytes([0x2F, 0x00, 0x00, 0x00, 0x00, 0x88, 0x88, 0x08])  # 1 x 1, no alpha -- minimal VP8L bitstream
    def chunk(tag, body):
        return tag + struct.pack("<I", len(body)) + body + (b"\x00" if len(body) & 1 else b"")
    vp8x = bytes([0x02, 0, 0, 0]) + struct.pack("<I", w - 1)[:3] + struct.pack("<I", h - 1)[:3]
    anim = struct.pack("<IH", 0, 0)
    anmf = struct.pack("<I", 0)[:3] * 2 + struct.pack("<I", 0)[:3] * 2 + struct.pack("<I", 100)[:3] + b"\x00" + chunk(b"VP8L", vp8l)
    body = b"WEBP" + chunk(b"VP8X", vp8x) + chunk(b"ANIM", anim) + chunk(b"ANMF", anmf)
    return b"RIFF" + struct.pack("<I", len(body)) + body


def test_webp_canvas_size_is_untrusted(hip_lib, small_set):
    """A ~100-byte animated WebP that claims a 16 000 x 16 000 canvas: webp_decoder_create must not allocate (let alone touch) a
    canvas-sized buffer, and the batch answers ErrBufTooSmall for the item while its neighbours are served."""
    import resource

    import lilliput_amd as la

    datas, exp = small_set
    evil = _vp8x_anim_claiming(16000, 16000)
    b = la.Batch(0)
    try:
        rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
        r = b.transform([datas[0], evil, datas[1]] + [evil] * 16, 64, 64, quality=85)
        rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
        assert r[0].data == exp[0] and r[2].data == exp[1]
        assert all(x.status in (1, 3) for x in r[3:]) and r[1].status in (1, 3), [x.status for x in r]  # ErrBufTooSmall (or rejected as invalid), never served, never a crash
        assert rss1 - rss0 < (512 << 10), (rss0, rss1)  # KiB: 17 x 1 GB canvases would show
    finally:
        b.close()


@pytest.mark.gpu
def test_transform_one_from_many_threads_shares_launches_and_keeps_the_bytes(hip_lib, oracle, fixture_bytes):
    """lilliput_hip_transform_one (Part B; lp_coalesce.h): 48 threads, each one image at a time through the process-wide dispatchers --
    what a Go service with one ImageOps per goroutine does (README.md:82-85). Every answer is the reference CPU path's bytes (integer
    scale) or passes the frame-level rule; errors come back per item."""
    import lilliput_amd as la

    names = ["large-sunrise.jpg", "coast.jpg", "field.jpg", "sunrise.jpg", "firefox-gray.jpg"]
    want = {n: oracle.transform_jpeg_thumbnail(fixture_bytes[n], 50, 50, 85) for n in names}
    outs, errs = {}, []

    def work(i):
        try:
            for k in range(3):
                n = names[(i + k) % len(names)]
                outs[(i, k)] = (n, la.transform_one(fixture_bytes[n], 50, 50, quality=85))
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(i,)) for i in range(48)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs[:3]
    assert len(outs) == 48 * 3
    ops = la.ImageOps(2048)
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    seen = {}
    for (i, k), (n, out) in outs.items():
        if out != want[n] and (n, out) not in seen:
            seen[(n, out)] = bench.firehose_check(la, oracle, ops, fixture_bytes[n], out, 50, 85)
            assert seen[(n, out)], n
    ops.Close()
    with pytest.raises(la.LilliputError):
        la.transform_one(b"\xff\xd8\xff\xe0 not a jpeg", 50, 50)


def test_resident_form_upload_run_download(hip_lib, small_set):
    """The resident form (lilliput_hip_batch_upload / _run / _download: compressed bytes in HBM, what bench.py --resident times): a set of
    600 items takes eight engines since round 6 (lp_batch.cpp batch_streams) and every part is cut into equal launches; one engine with an
    explicit launch size, and the default, must both hand back the reference's bytes for every item (a truncated file, which the reference's decoder refuses, among them)."""
    import lilliput_amd as la

    datas, exp = small_set
    n = 600
    srcs = [datas[i % len(datas)] for i in range(n)]
    want = [exp[i % len(datas)] for i in range(n)]
    srcs[17] = datas[3][: len(datas[3]) // 2]          # runs out of bytes: the reference's decoder fails
    want[17] = None
    b = la.Batch(0)
    for streams, chunk in ((0, 0), (1, 37), (3, 0)):
        b.upload(srcs, dst_cap=64 << 10, streams=streams)
        b.run(64, 64, la.ImageOpsFit, False, 85, chunk)
        res = b.download()
        assert len(res) == n
        for i, (r, w) in enumerate(zip(res, want)):
            if w is None:
                assert r.status != 0, i
            else:
                assert r.status == 0 and r.data == w, (streams, chunk, i)
    b.close()
