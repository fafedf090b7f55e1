#!/usr/bin/env python3
"""Does splitting the cfg2 batch over two HIP streams (each decoding half the captions) raise throughput?
Two host threads enqueue concurrently (ctypes releases the GIL during the C call)."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-captioning_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import torch
from gpu_util import plain_decoder
from capdec.engine import tf_greedy_ids

torch.manual_seed(0)
dec = plain_decoder(512, 10000, 4, 2).cuda().eval()
enc = torch.randn(1024, 1, 512).cuda()
N = 10


def loop(e, stream, n):
    with torch.cuda.stream(stream):
        for _ in range(n):
            tf_greedy_ids(dec, e, 20, 1, 2, resolve_dirty=False, to_cpu=False)


def timed(parts):
    streams = [torch.cuda.Stream() for _ in parts]
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=loop, args=(e, s, N if rep else 2)) for e, s in zip(parts, streams)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return 1024 * N / dt


print("one stream, B=1024        : %.0f captions/s" % timed([enc]))
print("two streams, B=512 each   : %.0f captions/s" % timed([enc[:512].contiguous(), enc[512:].contiguous()]))
print("four streams, B=256 each  : %.0f captions/s" % timed([enc[i * 256:(i + 1) * 256].contiguous() for i in range(4)]))
print("two streams, B=1024 each  : %.0f captions/s" % (2 * timed([enc, enc.clone()])))
