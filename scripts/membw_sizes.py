#!/usr/bin/env python3
"""Streaming ceilings (read, copy, triad; edigpu_membw) for buffers from L2-sized to HBM-sized: what a pass over a
vector of config 2 (94 MB, Infinity-Cache resident) can reach at best, next to the 1 GiB figures bench.py prints."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from edipack_amd import capi  # noqa: E402

capi.init(0)
for mb in (16, 32, 64, 94, 128, 192, 256, 512, 1024):
    rd, cp, tr = capi.membw(mb << 20)
    print(f"{mb:5d} MB per buffer: read {rd:7.0f}  copy {cp:7.0f}  triad {tr:7.0f} GB/s")
