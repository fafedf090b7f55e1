#!/usr/bin/env python3
"""A short run of every MFMA kernel family at the benchmark shapes, for the rocprofv3 passes of tools/mfma_util.py:
cfg2 greedy decode (vocab_reduce_kernel, the 1024 x 512 x 512 projections), cfg3 training (the 10240-row family and
the vocabulary-sized contractions), CNN-RNN greedy decode at B=32 and B=1024 (lstm_step_kernel), the cfg5 head."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "image-captioning_amd")):
    sys.path.insert(0, p)
import torch

import bench
from capdec import engine
from capdec.trainer import CaptionTrainer
from models.cnnrnn_model import CNNtoRNN
from models.vitattn_model import VITAttentionModel
from models.yolocnnattn_model import YOLOCNNAttentionModel

dev = torch.device("cuda", 0)
E, H, V, B, LEN = bench.E, bench.H, bench.V, bench.B, bench.LEN
g = torch.Generator().manual_seed(1)
with torch.no_grad():
    m = bench.build_model(dev)
    feats = torch.rand(B, bench.D, generator=g).to(dev)
    for _ in range(2):
        engine.tf_greedy_ids(m, m.encode_features(feats).view(B, 1, E), LEN, 1, 2, to_cpu=True)
    for Br in (32, 1024):
        r = CNNtoRNN(256, 512, 5000).to(dev).eval()
        fr = torch.rand(Br, 2048, generator=g).to(dev)
        for _ in range(2):
            r.caption_ids(fr, max_length=LEN)
    y = YOLOCNNAttentionModel(E, V, H, 2).to(dev).eval()
    fy = torch.rand(B, 88484, generator=g).to(dev)
    for _ in range(2):
        y.encode_features(fy)
torch.manual_seed(0)
t = VITAttentionModel(E, V, H, 3).to(dev)
tr = CaptionTrainer(t, lr=1e-3, wgrad_stream=False)      # one stream: per-kernel durations not inflated by overlap
ft = torch.rand(512, 1000, generator=g).to(dev)
caps = bench.synthetic_captions(g, 512, 20).to(dev)
for _ in range(3):
    tr.step(ft, caps)
torch.cuda.synchronize()
print("done")
