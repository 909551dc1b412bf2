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


def synth_jpeg(seed, size=4096, quality=90, restart_rows=0, width=None, height=None):
    """Returns the JPEG bytes of synthetic image `seed` (optionally cropped to width x height)."""
    from PIL import Image

    rgb = synth_rgb(seed, size)
    if width or height:
        rgb = np.ascontiguousarray(rgb[: (height or size), : (width or size)])
    b = io.BytesIO()
    kw = {}
    if restart_rows:
        kw["restart_marker_rows"] = restart_rows
    Image.fromarray(rgb).save(b, "JPEG", quality=quality, subsampling=2, optimize=False, **kw)
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
    else:
        h, w = rgb.shape[:2]
        return b"LPPIXELS" + struct.pack("<6I", w, h, 3, 0, 1 + (seed % 8 if seed % 5 == 0 else 0), 0) + np.ascontiguousarray(rgb[..., ::-1]).tobytes()
    return b.getvalue()


def _fh_job(args):
    return firehose_source(*args)


def firehose_pool(distinct_per_kind, lo=512, hi=4096, seed=1, workers=None):
    """{kind: [bytes]}: `distinct_per_kind` sources per format, sides log-uniform in [lo, hi] (seeded), generated on the host cores."""
    import multiprocessing as mp
    import os

    rng = np.random.default_rng(seed)
    jobs = []
    for k, (kind, _) in enumerate(FIREHOSE_MIX):
        for i in range(distinct_per_kind):
            side = int(round(float(np.exp(rng.uniform(np.log(lo), np.log(hi)))) / 8) * 8)
            jobs.append((kind, 1000 * (k + 1) + i, side))
    workers = workers or max(1, min(len(jobs), (os.cpu_count() or 2) - 1, 64))
    if workers <= 1:
        out = [firehose_source(*j) for j in jobs]
    else:
        with mp.get_context("fork").Pool(workers) as pool:
            out = pool.map(_fh_job, jobs, chunksize=1)
    pools = {kind: [] for kind, _ in FIREHOSE_MIX}
    for j, d in zip(jobs, out):
        pools[j[0]].append(d)
    return pools


def firehose_items(pools, n, seed=2):
    """n items drawn from the pools with the mix's probabilities: [(kind, bytes)]."""
    rng = np.random.default_rng(seed)
    kinds = rng.choice([k for k, _ in FIREHOSE_MIX], size=n, p=[p for _, p in FIREHOSE_MIX])
    return [(str(k), pools[str(k)][int(rng.integers(len(pools[str(k)])))]) for k in kinds]
