"""Concurrency check: N host threads, each with its own Decoder / ImageOps (one per goroutine in the reference) and some with
their own Batch, hammer the same GPU; every output must equal the single-threaded answer."""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lilliput_amd as la
from oracle import oracle as O
import gif_cases, png_cases
fx = os.path.join(ROOT, "tests", "golden", "inputs")
srcs = [open(os.path.join(fx, n), "rb").read() for n in sorted(os.listdir(fx))]
srcs += [gif_cases.fixtures()["party-discord.gif"], png_cases.fixtures()["firefox.png"], png_cases.fixtures()["ferry_sunset.png"]]
def one(ops, data, w, h):
    d = la.Decoder(data)
    try:
        return ops.Transform(d, la.ImageOptions(".jpeg", w, h, la.ImageOpsFit, False, {la.JpegQuality: 80}, EncodeTimeout=10**11))
    finally:
        d.Close()
ops0 = la.ImageOps(2048)
expect = {(k, w): one(ops0, s, w, w) for k, s in enumerate(srcs) for w in (24, 64, 100)}
errors = []
def worker(tid, iters):
    rng = np.random.default_rng(tid)
    ops = la.ImageOps(2048)
    batch = la.Batch(0) if tid % 2 == 0 else None
    for it in range(iters):
        k, w = int(rng.integers(len(srcs))), int(rng.choice([24, 64, 100]))
        if batch is not None and it % 3 == 0:
            ks = [int(x) for x in rng.integers(0, len(srcs), 6)]
            res = batch.transform([srcs[q] for q in ks], w, w, quality=80)
            for q, r in zip(ks, res):
                if r.status != 0 or r.data != expect[(q, w)]: errors.append(("batch", tid, q, w, r.status))
        else:
            if one(ops, srcs[k], w, w) != expect[(k, w)]: errors.append(("one", tid, k, w))
    ops.Close()
    if batch: batch.close()
t0 = time.time()
th = [threading.Thread(target=worker, args=(t, 150)) for t in range(12)]
[t.start() for t in th]; [t.join() for t in th]
print("12 threads x 150 iterations: errors =", len(errors), errors[:5], "%.1fs" % (time.time() - t0))
