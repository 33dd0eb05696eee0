#!/usr/bin/env python3
"""Host-to-device rate of the frame upload path (row f2), no decode: pinned staging buffers -> tl3d_upload_frame_async, 1080x1920.
Prints GB/s for f32 depth + BGR, u16 depth + BGR, depth alone; and torch's own pinned copy for reference."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import tl3d  # noqa: E402
from tl3d.fusion import PinnedArray  # noqa: E402

W, H = 1080, 1920
N = 64
ctx = tl3d.FusionContext(W, H, 1719.0, 1719.0, 540.0, 960.0, n_slots=N, grid=None)
S = 8
f32 = [PinnedArray((H, W), np.float32) for _ in range(S)]
u16 = [PinnedArray((H, W), np.uint16) for _ in range(S)]
bgr = [PinnedArray((H, W, 3), np.uint8) for _ in range(S)]
for b in range(S):
    f32[b].array[:] = 1.0
    u16[b].array[:] = 1000
    bgr[b].array[:] = 7


def run(name, fn, nbytes, reps=3):
    fn()
    ctx.sync()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ctx.sync()
        best = min(best, time.perf_counter() - t0)
    print(f"{name:52s} {1e3 * best / N:7.3f} ms/frame  {N * nbytes / best / 1e9:6.1f} GB/s  {N / best:8.1f} frames/s", flush=True)


run("f32 depth + BGR (14.5 MB)", lambda: [ctx.upload_async(i, f32[i % S].array, bgr[i % S].array) for i in range(N)], H * W * 7)
run("u16 depth + BGR (10.4 MB)", lambda: [ctx.upload_async(i, u16[i % S].array, bgr[i % S].array) for i in range(N)], H * W * 5)
run("f32 depth alone (8.3 MB)", lambda: [ctx.upload_async(i, f32[i % S].array, None) for i in range(N)], H * W * 4)
run("u16 depth alone (4.1 MB)", lambda: [ctx.upload_async(i, u16[i % S].array, None) for i in range(N)], H * W * 2)
dev = torch.device("cuda", 0)
src = [torch.empty((H, W), dtype=torch.float32).pin_memory() for _ in range(S)]
dst = [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(S)]
streams = [torch.cuda.Stream() for _ in range(4)]
for ns in (1, 2, 4):
    def go():
        for i in range(N):
            with torch.cuda.stream(streams[i % ns]):
                dst[i % S].copy_(src[i % S], non_blocking=True)
        torch.cuda.synchronize()
    go()
    t0 = time.perf_counter()
    go()
    t = time.perf_counter() - t0
    print(f"torch pinned f32 copy, {ns} stream(s)                      {1e3 * t / N:7.3f} ms/frame  {N * H * W * 4 / t / 1e9:6.1f} GB/s", flush=True)
