#!/bin/bash
for dt in f32 bf16; do
  timeout -k 10 120 python tools/upsample_probe.py 32 80 96 80 2 $dt 2>/dev/null || exit 1
  timeout -k 10 120 python tools/upsample_probe.py 64 40 48 40 2 $dt 2>/dev/null || exit 1
  timeout -k 10 120 python tools/upsample_probe.py 8 16 16 16 64 $dt 2>/dev/null || exit 1
done
