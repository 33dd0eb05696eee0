#!/usr/bin/env python3
"""How many HIP streams really run side by side in this process?  n one-wave kernels of `spin` ms on n fresh streams take
ceil(n / Q) * spin when the runtime multiplexes streams onto Q hardware queues (tl3d_probe_hw_queues).
    GPU_MAX_HW_QUEUES=4 python tools/probe_queues.py      # ROCm default
    python tools/probe_queues.py                          # the package default (24)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tl3d  # noqa: E402,F401  (sets GPU_MAX_HW_QUEUES before the first HIP call)
from tl3d import _cabi as abi  # noqa: E402

ns = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 24, 32, 48]
for n in ns:
    for spin in (0.25, 1.0, 0.25):
        print(json.dumps(abi.probe_hw_queues(0, n, spin)))
