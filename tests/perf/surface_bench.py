"""Average surface distance of two 160x192x160 masks: device path vs the scipy oracle on one host core (the reference's
validation spends 5-7 s per volume here, results_validation.ipynb:267).   python tests/perf/surface_bench.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mri_epilepsy_diagnosis_amd.segmentation import surface  # noqa: E402
from oracle import metrics as O_MET  # noqa: E402

gt, pred = O_MET.seeded_blobs(5, (160, 192, 160))
g, p = torch.from_numpy(gt).cuda(), torch.from_numpy(pred).cuda()
surface.average_surface_distance(g, p)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    got = surface.average_surface_distance(g, p)
gpu_ms = (time.perf_counter() - t0) / 5 * 1e3
area = np.load(os.path.join(ROOT, "mri_epilepsy_diagnosis_amd", "segmentation", "data", "surfel_area_spacing111.npy"))
t0 = time.perf_counter()
ref = O_MET.average_surface_distance(gt, pred, area)
cpu_ms = (time.perf_counter() - t0) * 1e3
print("ASD 160x192x160: device %.2f ms/volume, scipy oracle on 1 core %.0f ms/volume; device %s oracle %s"
      % (gpu_ms, cpu_ms, got, ref))
