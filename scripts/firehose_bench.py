"""BASELINE configs[4] in miniature: a mixed-format firehose (JPEG / PNG / WebP, 512-4096 px, a handed-over decoded frame standing in for
the AVIF / video items) -> 256 px JPEG through ONE lilliput_hip_batch_transform call; prints images/s and the per-format split.
  python scripts/firehose_bench.py [n_items]            (through gpurun)
"""
import io
import os
import struct
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lilliput_amd as la  # noqa: E402
from lilliput_amd import synth  # noqa: E402


def main():
    from PIL import Image

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    rng = np.random.default_rng(1)
    pool = {"jpeg": [], "png": [], "webp": [], "pixels": []}
    t0 = time.time()
    for k, size in enumerate((512, 768, 1024, 1536, 2048, 3072, 4096)):
        rgb = synth.synth_rgb(100 + k, size)
        im = Image.fromarray(rgb)
        b = io.BytesIO(); im.save(b, "JPEG", quality=88, subsampling=2); pool["jpeg"].append(b.getvalue())
        if size <= 2048:
            b = io.BytesIO(); im.save(b, "PNG", compress_level=3); pool["png"].append(b.getvalue())
            b = io.BytesIO(); im.save(b, "WEBP", quality=80); pool["webp"].append(b.getvalue())
        if size <= 1024:
            bgr = np.ascontiguousarray(rgb[..., ::-1])
            pool["pixels"].append(b"LPPIXELS" + struct.pack("<6I", size, size, 3, 0, 1, 0) + bgr.tobytes())
    kinds = rng.choice(["jpeg", "png", "webp", "pixels"], size=n, p=[0.6, 0.15, 0.15, 0.1])
    items = [pool[k][int(rng.integers(len(pool[k])))] for k in kinds]
    print("sources ready in %.1fs: %s" % (time.time() - t0, {k: int((kinds == k).sum()) for k in pool}), flush=True)
    b = la.Batch(0)
    b.transform(items[:64], 256, 256, quality=85)  # warm-up: engines, arenas, worker pool
    t0 = time.time()
    res = b.transform(items, 256, 256, quality=85)
    dt = time.time() - t0
    b.close()
    ok = sum(1 for r in res if r.status == 0)
    mb = sum(len(x) for x in items) / 1e6
    print("%d items (%d ok) in %.3fs = %.0f images/s, %.1f MB of sources (%.2f GB/s)" % (n, ok, dt, n / dt, mb, mb / dt / 1e3))


if __name__ == "__main__":
    main()
