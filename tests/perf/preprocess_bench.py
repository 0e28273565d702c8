"""Histogram standardisation of one 160x192x160 volume: device path vs the numpy oracle on one host core (the reference's
collate function runs it per sample on the CPU).   python tests/perf/preprocess_bench.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mri_epilepsy_diagnosis_amd.classification import preprocessing as P  # noqa: E402
from oracle import preprocessing as O_PRE  # noqa: E402

lm = np.array([0.0, 4.5, 11.0, 14.2, 17.9, 26.0, 35.5, 47.0, 58.0, 63.1, 69.0, 84.0, 100.0])
vol = O_PRE.synthetic_t1(1, (160, 192, 160))
xd = torch.from_numpy(vol).cuda()
P.normalize(xd, lm)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    out = P.normalize(xd, lm)
torch.cuda.synchronize()
gpu_ms = (time.perf_counter() - t0) / 10 * 1e3
t0 = time.perf_counter()
ref = O_PRE.normalize(vol, lm)
cpu_ms = (time.perf_counter() - t0) * 1e3
print("hist-std 160x192x160: device %.2f ms/volume (incl. the 26-float D2H sync), numpy on 1 core %.0f ms/volume, identical=%s"
      % (gpu_ms, cpu_ms, np.array_equal(out.cpu().numpy(), ref)))
