"""Synthetic workload of BASELINE.md section 3: photo-like 8-bit RGB images (low-frequency sinusoid gradients
+ bicubic-upsampled 128x128 Gaussian noise (sigma 40) + per-pixel Gaussian noise (sigma 6)), seeded per image
with numpy.random.default_rng(i), encoded once on the host as baseline 4:2:0 q90 JPEG with the Annex-K
Huffman tables and no restart markers. Used by bench.py and the tests; not part of the product path."""
import io

import numpy as np


def synth_rgb(seed, size=4096):
    from PIL import Image

    rng = np.random.default_rng(seed)
    t = np.arange(size, dtype=np.float32) / size
    img = np.empty((size, size, 3), np.float32)
    tau = np.float32(2 * np.pi)
    for c in range(3):
        f = rng.uniform(0.5, 3.0, 4).astype(np.float32)
        ph = rng.uniform(0, 2 * np.pi, 4).astype(np.float32)
        # 128 + 50 sin(2 pi f0 x + ph0) cos(2 pi f1 y + ph1) + 30 sin(2 pi (f2 x + f3 y) + ph2): both terms are sums of outer products
        # of per-axis vectors (sin(a + b) = sin a cos b + cos a sin b), so the gradients cost O(size) trigonometry, not O(size^2)
        sx, cy = np.sin(tau * f[0] * t + ph[0]), np.cos(tau * f[1] * t + ph[1])
        ax, by = tau * f[2] * t + ph[2], tau * f[3] * t
        plane = img[:, :, c]
        np.multiply.outer(np.float32(50) * cy, sx, out=plane)
        plane += np.multiply.outer(np.float32(30) * np.cos(by), np.sin(ax))
        plane += np.multiply.outer(np.float32(30) * np.sin(by), np.cos(ax))
        lo = rng.normal(0, 40, (128, 128)).astype(np.float32)
        plane += np.asarray(Image.fromarray(lo, mode="F").resize((size, size), Image.BICUBIC))
        noise = rng.standard_normal((size, size), dtype=np.float32)
        noise *= np.float32(6)
        plane += noise
    img += np.float32(128.5)
    return np.clip(img, 0, 255, out=img).astype(np.uint8)


def synth_jpeg(seed, size=4096, quality=90, restart_rows=0, width=None, height=None, subsampling=2, progressive=False):
    """Returns the JPEG bytes of synthetic image `seed` (optionally cropped to width x height)."""
    from PIL import Image

    rgb = synth_rgb(seed, size)
    if width or height:
        rgb = np.ascontiguousarray(rgb[: (height or size), : (width or size)])
    b = io.BytesIO()
    kw = {}
    if restart_rows:
        kw["restart_marker_rows"] = restart_rows
    if progressive:
        kw["progressive"] = True   # libjpeg default script: ten scans
    Image.fromarray(rgb).save(b, "JPEG", quality=quality, subsampling=subsampling, optimize=False, **kw)  # Pillow: 2 = 4:2:0, 1 = 4:2:2, 0 = 4:4:4
    return b.getvalue()


def _job(args):
    return synth_jpeg(*args)


def synth_jpeg_set(n, size=4096, quality=90, workers=None):
    """n distinct synthetic JPEGs (seeds 0..n-1), generated in parallel on the host cores."""
    import multiprocessing as mp
    import os

    workers = workers or min(n, max(1, (os.cpu_count() or 2) - 1))
    if workers <= 1 or n <= 1:
        return [synth_jpeg(i, size, quality) for i in range(n)]
    with mp.get_context("fork").Pool(workers) as pool:
        return pool.map(_job, [(i, size, quality) for i in range(n)])


# ---- BASELINE configs[4] in miniature: the mixed-format firehose (JPEG 70 %, PNG 15 %, WebP 10 %, AVIF 5 %;
# side lengths log-uniform). AV1 is a host codec outside this library: its items arrive as handed-over decoded frames
# (include/lilliput_hip.h lilliput_hip_pixels_header), which is how a service with libavif in front would feed them.
FIREHOSE_MIX = (("jpeg", 0.70), ("png", 0.15), ("webp", 0.10), ("pixels", 0.05))
# The same law with REAL AVIF files in the 5 % share (bench.py --workload firehose when Pillow can write AVIF): the AV1 decode then runs in
# the bench's host feeder (avif_to_handover below: Pillow's bundled libavif), inside the timed region, and its frames enter the library
# through the same hand-over item.
FIREHOSE_MIX_AVIF = (("jpeg", 0.70), ("png", 0.15), ("webp", 0.10), ("avif", 0.05))


def avif_supported():
    try:
        from PIL import features

        return bool(features.check("avif"))
    except Exception:
        return False


# ---- the bench's AVIF feeder as worker PROCESSES (Pillow's decoder holds the interpreter lock: threads do not scale). Workers are spawned
# (no fork of a process that has the GPU runtime up), attach to one shared-memory block and write each decoded frame -- hand-over header
# + BGR(A) rows -- at the offset the parent names; the parent wraps the block's slices as the items of its transform call.
_feed_shm = None


def _avif_feed_init(shm_name):
    global _feed_shm
    from multiprocessing import shared_memory

    from PIL import AvifImagePlugin

    AvifImagePlugin.DEFAULT_MAX_THREADS = 1  # one decoder thread per worker: the workers are the parallelism
    _feed_shm = shared_memory.SharedMemory(name=shm_name)


def _avif_feed_job(job):
    data, off = job
    frame = avif_to_handover(data)
    np.frombuffer(_feed_shm.buf, dtype=np.uint8, count=frame.size, offset=off)[:] = frame
    return frame.size


def avif_frame_bytes(data):
    """Size of the hand-over item avif_to_handover(data) produces (header + pixels), from the file's header alone."""
    from PIL import Image

    im = Image.open(io.BytesIO(bytes(data)))
    return 32 + im.size[0] * im.size[1] * (4 if im.mode in ("RGBA", "LA", "PA") else 3)


class AvifFeeder:
    """`workers` spawned processes that decode AVIF files into a shared-memory block of `capacity` bytes."""

    def __init__(self, workers, capacity):
        import multiprocessing as mp
        from multiprocessing import shared_memory

        self.shm = shared_memory.SharedMemory(create=True, size=max(1, int(capacity)))
        self.pool = mp.get_context("spawn").Pool(int(workers), initializer=_avif_feed_init, initargs=(self.shm.name,))
        self.pool.map(_avif_noop, range(int(workers) * 2))  # the workers are up (interpreter + Pillow imported) before anything is timed

    def submit(self, files, sizes):
        """Start decoding `files` (sizes[i] = avif_frame_bytes(files[i])); returns a handle for collect()."""
        offs, o = [], 0
        for n in sizes:
            offs.append(o)
            o += (n + 63) // 64 * 64
        assert o <= self.shm.size
        return self.pool.map_async(_avif_feed_job, list(zip(files, offs)), chunksize=1), offs, sizes

    def collect(self, handle):
        res, offs, sizes = handle
        got = res.get()
        assert list(got) == list(sizes)
        return [np.frombuffer(self.shm.buf, dtype=np.uint8, count=n, offset=o) for o, n in zip(offs, sizes)]

    def close(self):
        self.pool.close()
        self.pool.join()
        self.shm.close()
        self.shm.unlink()


def _avif_noop(_):
    return 0


def avif_to_handover(data):
    """What a service with an AV1 decoder in front of the library does (lilliput.go:136-164 -> avif.cpp:277-321 leaves BGR(A) in the
    framebuffer): decode the AVIF file on the host -- here with Pillow's bundled libavif -- and wrap the frame as a hand-over item
    (include/lilliput_hip.h lilliput_hip_pixels_header). Returns a numpy uint8 array."""
    import struct

    from PIL import Image

    im = Image.open(io.BytesIO(bytes(data)))
    im.load()
    has_alpha = im.mode in ("RGBA", "LA", "PA")
    a = np.asarray(im.convert("RGBA" if has_alpha else "RGB"))
    bgr = a[..., [2, 1, 0, 3]] if has_alpha else a[..., ::-1]
    h, w, cn = bgr.shape
    out = np.empty(32 + h * w * cn, dtype=np.uint8)
    out[:32] = np.frombuffer(b"LPPIXELS" + struct.pack("<6I", w, h, cn, 0, 1, 0), dtype=np.uint8)
    out[32:] = np.ascontiguousarray(bgr).reshape(-1)
    return out


def firehose_source(kind, seed, side):
    """One source of the mix: `kind` in jpeg / png / webp / pixels, content = synth_rgb(seed, side) (aspect 4:3 for odd seeds)."""
    import struct

    from PIL import Image

    rgb = synth_rgb(seed, side)
    if seed & 1:
        rgb = np.ascontiguousarray(rgb[: max(8, side * 3 // 4)])
    im = Image.fromarray(rgb)
    b = io.BytesIO()
    if kind == "jpeg":
        im.save(b, "JPEG", quality=(75, 85, 90, 95)[seed % 4], subsampling=(2, 2, 1, 0)[seed % 4], optimize=bool(seed % 3 == 0))
    elif kind == "png":
        if seed % 4 == 3:
            im = im.convert("RGBA")
        im.save(b, "PNG", compress_level=3)
    elif kind == "webp":
        im.save(b, "WEBP", quality=80, method=2) if seed % 4 else im.save(b, "WEBP", lossless=True, method=0)
    elif kind == "avif":
        im.save(b, "AVIF", quality=(60, 75, 85)[seed % 3], speed=8, subsampling=("4:2:0", "4:4:4")[seed % 2])
    else:
        h, w = rgb.shape[:2]
        return b"LPPIXELS" + struct.pack("<6I", w, h, 3, 0, 1 + (seed % 8 if seed % 5 == 0 else 0), 0) + np.ascontiguousarray(rgb[..., ::-1]).tobytes()
    return b.getvalue()


def _fh_job(args):
    return firehose_source(*args)


def firehose_pool(distinct_per_kind, lo=512, hi=4096, seed=1, workers=None, mix=None):
    """{kind: [bytes]}: `distinct_per_kind` sources per format, sides log-uniform in [lo, hi] (seeded), generated on the host cores."""
    import multiprocessing as mp
    import os

    mix = mix or FIREHOSE_MIX
    rng = np.random.default_rng(seed)
    jobs = []
    for k, (kind, _) in enumerate(mix):
        for i in range(distinct_per_kind):
            side = int(round(float(np.exp(rng.uniform(np.log(lo), np.log(hi)))) / 8) * 8)
            jobs.append((kind, 1000 * (k + 1) + i, side))
    workers = workers or max(1, min(len(jobs), (os.cpu_count() or 2) - 1, 64))
    if workers <= 1:
        out = [firehose_source(*j) for j in jobs]
    else:
        with mp.get_context("fork").Pool(workers) as pool:
            out = pool.map(_fh_job, jobs, chunksize=1)
    pools = {kind: [] for kind, _ in mix}
    for j, d in zip(jobs, out):
        pools[j[0]].append(d)
    return pools


def firehose_items(pools, n, seed=2, mix=None):
    """n items drawn from the pools with the mix's probabilities: [(kind, bytes)]."""
    mix = mix or FIREHOSE_MIX
    rng = np.random.default_rng(seed)
    kinds = rng.choice([k for k, _ in mix], size=n, p=[p for _, p in mix])
    return [(str(k), pools[str(k)][int(rng.integers(len(pools[str(k)])))]) for k in kinds]
