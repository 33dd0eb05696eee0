"""File formats either side of the hot path, cv2-free.

Loader rules follow depth_to_reconstruction.py:80-119 and :439-477; the depth producer whose
files these are is depth_processor.py:905-921 (`<id>_depth.npy`, `<id>_depth.png` u16 mm).
PLY writers follow depth_to_reconstruction.py:673-703 / depth_enhanced_reconstruction.py:1283-1311.
OpenCV is not installed in the build image: images are decoded with PIL, and the bilinear resize
restates cv2.INTER_LINEAR (half-pixel centres, replicated border) in numpy -- that restatement is
parity-unpinned (no cv2 to check it against, no reference fixture).
"""
from __future__ import annotations

from pathlib import Path
from typing import List, Optional, Tuple

import numpy as np

IMAGE_SUFFIXES = (".png", ".jpg", ".jpeg")


# ---------------------------------------------------------------------------------------------
# depth files
# ---------------------------------------------------------------------------------------------
class DepthImageLoader:
    """Same static interface as the reference class (D2R:76-119)."""

    @staticmethod
    def load_depth(filepath, raw_u16: bool = False) -> Optional[np.ndarray]:
        """.npy -> float32 as stored; .png (16-bit) -> millimetres / 1000 -> metres (D2R:82-90).

        raw_u16=True returns the uint16 millimetre image untouched so the device does the /1000
        (tl3d_upload_frame, TL3D_DEPTH_U16_MM) -- bit-identical to the host conversion.
        """
        filepath = Path(filepath)
        if filepath.suffix == ".npy":
            return np.load(str(filepath)).astype(np.float32, copy=False)
        if filepath.suffix == ".png":
            from PIL import Image
            try:
                with Image.open(str(filepath)) as im:
                    if im.mode in ("I;16", "I;16L"):             # the producer's format (DP:905-921): one 2-byte pack, no object protocol
                        w_, h_ = im.size
                        arr = np.frombuffer(im.tobytes(), np.uint16).reshape(h_, w_)
                    else:
                        arr = np.array(im)
            except Exception:
                return None
            if arr.ndim == 3:
                # cv2.imread(path, IMREAD_ANYDEPTH) (D2R:87) has no ANYCOLOR flag: a colour PNG comes back as ONE gray channel,
                # 0.299 R + 0.587 G + 0.114 B (alpha dropped) at the file's bit depth.  Restated, not pinned (no cv2 here): libpng
                # does that sum in fixed point, so a value may differ by 1 from this rounding.
                rgb = arr[..., :3].astype(np.float64)
                gray = np.rint(0.299 * rgb[..., 0] + 0.587 * rgb[..., 1] + 0.114 * rgb[..., 2])
                arr = gray.astype(arr.dtype)
            if raw_u16 and arr.dtype == np.uint16:
                return arr
            return arr.astype(np.float32) / 1000.0
        if filepath.suffix in (".exr", ".EXR"):
            # the reference reads EXR through OpenCV (cv2.imread(..., IMREAD_ANYDEPTH) -> one float32 channel, D2R:92-95)
            try:
                return read_exr_depth(filepath)
            except (OSError, ValueError) as e:
                print(f"  Warning: cannot read EXR depth {filepath.name}: {e}")
                return None
        return None

    @staticmethod
    def find_matching_depth(rgb_name: str, depth_folder) -> Optional[Path]:
        """First existing of <stem>_depth.npy, <stem>_depth.png, <stem>.npy, <stem>.png, depth_<stem>.npy,
        depth_<stem>.png (D2R:105-112)."""
        depth_folder = Path(depth_folder)
        stem = Path(rgb_name).stem
        for name in (f"{stem}_depth.npy", f"{stem}_depth.png", f"{stem}.npy", f"{stem}.png",
                     f"depth_{stem}.npy", f"depth_{stem}.png"):
            cand = depth_folder / name
            if cand.exists():
                return cand
        return None


def read_exr_depth(path) -> np.ndarray:
    """Minimal OpenEXR reader for depth maps: single-part scan-line files, compression NONE / ZIPS / ZIP, HALF / FLOAT / UINT
    channels.  Returns ONE float32 channel as `cv2.imread(path, cv2.IMREAD_ANYDEPTH)` does for the reference (D2R:92-95):
    the only channel, else `Z`, else `Y`, else the grey value 0.299 R + 0.587 G + 0.114 B.  Tiled, multi-part, deep and
    PIZ / PXR24 / B44 / DWA files raise ValueError (OpenCV is not available in this image: written from the OpenEXR file
    layout, parity unpinned)."""
    import struct
    import zlib
    with open(str(path), "rb") as f:
        data = f.read()
    if len(data) < 8 or struct.unpack_from("<I", data, 0)[0] != 20000630:
        raise ValueError("not an OpenEXR file")
    flags = struct.unpack_from("<I", data, 4)[0]
    if flags & 0x1A00:
        raise ValueError("tiled / deep / multi-part OpenEXR files are not supported")
    pos, attrs = 8, {}
    while data[pos] != 0:
        e = data.index(b"\0", pos); name = data[pos:e].decode("latin1"); pos = e + 1
        e = data.index(b"\0", pos); typ = data[pos:e].decode("latin1"); pos = e + 1
        size = struct.unpack_from("<i", data, pos)[0]; pos += 4
        attrs[name] = (typ, data[pos:pos + size]); pos += size
    pos += 1
    comp = attrs["compression"][1][0]
    if comp not in (0, 2, 3):
        raise ValueError(f"OpenEXR compression {comp} is not supported (NONE, ZIPS, ZIP are)")
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    chans, cp, cl = [], 0, attrs["channels"][1]
    while cl[cp] != 0:
        e = cl.index(b"\0", cp); cname = cl[cp:e].decode("latin1"); cp = e + 1
        ptype, _pl, xs, ys = struct.unpack_from("<iB3xii", cl, cp); cp += 16
        if xs != 1 or ys != 1:
            raise ValueError("sub-sampled OpenEXR channels are not supported")
        chans.append((cname, ptype))
    psize = {0: 4, 1: 2, 2: 4}
    row_bytes = sum(psize[t] * w for _, t in chans)
    lines = {0: 1, 2: 1, 3: 16}[comp]
    nchunks = (h + lines - 1) // lines
    offsets = struct.unpack_from(f"<{nchunks}Q", data, pos)
    planes = {c: np.zeros((h, w), np.float32) for c, _ in chans}
    for off in offsets:
        y, sz = struct.unpack_from("<ii", data, off)
        raw = data[off + 8:off + 8 + sz]
        nl = min(lines, y0 + h - y)
        want = row_bytes * nl
        if comp != 0 and sz < want:
            t = np.frombuffer(zlib.decompress(raw), np.uint8).astype(np.int64)
            if len(t) != want:
                raise ValueError("corrupt OpenEXR chunk")
            t = (np.cumsum(t - 128) + 128) % 256                     # undo the byte-delta predictor: t[i] += t[i-1] - 128
            half = (want + 1) // 2
            buf = np.empty(want, np.uint8)
            buf[0::2], buf[1::2] = t[:half], t[half:]                  # undo the even / odd byte split
            raw = buf.tobytes()
        p = 0
        for r in range(nl):
            for cname, ptype in chans:
                n = psize[ptype] * w
                seg = raw[p:p + n]; p += n
                dt = {0: "<u4", 1: "<f2", 2: "<f4"}[ptype]
                planes[cname][y - y0 + r] = np.frombuffer(seg, dt).astype(np.float32)
    names = [c for c, _ in chans]
    if len(names) == 1:
        return planes[names[0]]
    for pick in ("Z", "Y"):
        if pick in planes:
            return planes[pick]
    if all(c in planes for c in "RGB"):
        return (np.float32(0.299) * planes["R"] + np.float32(0.587) * planes["G"] + np.float32(0.114) * planes["B"]).astype(np.float32)
    return planes[names[0]]


def _pil_row_pointers(im, width: int, height: int, pixelsize: int):
    """Address of PIL's table of row pointers for a loaded image, or None.  im.getim() hands out the library's image
    struct (the capsule Tk / Qt bindings use); its layout is read defensively -- bands / xsize / ysize / pixelsize /
    linesize at the offsets of Pillow 11-12 must all agree with what the image says -- and anything else falls back to
    tobytes().  What it buys: the pixels never pass through a Python bytes object under the interpreter lock."""
    import ctypes
    try:
        import PIL
        # The struct is private to Pillow: only the releases whose layout was checked (12.x: mode id, type, depth, bands, xsize,
        # ysize as ints 0..5; pixelsize / linesize as ints 18, 19; row pointers at byte 48) take the fast path, every other
        # release decodes through tobytes().
        if not PIL.__version__.startswith("12."):
            return None
        cap = im.getim()
        api = ctypes.pythonapi
        api.PyCapsule_GetName.restype, api.PyCapsule_GetName.argtypes = ctypes.c_char_p, [ctypes.py_object]
        api.PyCapsule_GetPointer.restype, api.PyCapsule_GetPointer.argtypes = ctypes.c_void_p, [ctypes.py_object, ctypes.c_char_p]
        name = api.PyCapsule_GetName(cap)
        if name != b"Pillow Imaging":
            return None
        base = api.PyCapsule_GetPointer(cap, name)
        if not base:
            return None
        ints = (ctypes.c_int32 * 20).from_address(base)
        bands, xs, ys, px, ls = ints[3], ints[4], ints[5], ints[18], ints[19]
        if (xs, ys, px, ls) != (width, height, pixelsize, width * pixelsize) or bands != len(im.getbands()):
            return None
        rows = ctypes.c_void_p.from_address(base + 48).value
        return rows or None
    except Exception:
        return None


def decode_bgr_into(path, dst: np.ndarray) -> bool:
    """Decode an image file straight into dst (uint8 [H, W, 3], C-contiguous, e.g. pinned staging) in B, G, R order.
    False if the file cannot be read; ValueError if its size is not dst's."""
    from PIL import Image
    h, w = dst.shape[:2]
    try:
        with Image.open(str(path)) as im:
            if im.mode != "RGB":
                im = im.convert("RGB")
            else:
                im.load()                                  # the decoders release the interpreter lock
            if im.size != (w, h):
                raise ValueError(f"image {path} is {(im.size[1], im.size[0])}, expected {(h, w)}")
            rows = _pil_row_pointers(im, w, h, 4)
            if rows is not None:
                from . import _cabi as abi
                abi.check(abi.load().tl3d_host_pack_bgr_rows(dst.ctypes.data, rows, h, w))       # no lock held in here
            else:
                data = im.tobytes("raw", "BGR")
                copy_bytes(dst, data, len(data))
            return True
    except ValueError:
        raise
    except Exception:
        return False


def _image_bgr_bytes(path):
    """(bytes in B, G, R order, height, width) or None.  The pack to BGR happens inside PIL's raw encoder (2.6 ms per
    1080p frame); np.array(im.convert("RGB")) followed by a [..., ::-1] copy held the interpreter lock for 25 ms."""
    from PIL import Image
    try:
        with Image.open(str(path)) as im:
            if im.mode != "RGB":
                im = im.convert("RGB")            # alpha dropped, grey replicated (cv2.imread's default)
            w, h = im.size
            return im.tobytes("raw", "BGR"), h, w
    except Exception:
        return None


def read_image_bgr(path) -> Optional[np.ndarray]:
    """uint8 [H,W,3] in BGR order, what cv2.imread returns (alpha dropped, grey replicated)."""
    from PIL import Image
    try:
        with Image.open(str(path)) as im:
            w, h = im.size
        out = np.empty((h, w, 3), np.uint8)
        return out if decode_bgr_into(path, out) else None
    except Exception:
        return None


_pil_cache_blocks = 0


def keep_pil_blocks(n_images: int):
    """Let Pillow keep the storage of `n_images` decoded 1080p-class images for re-use (Image.core.set_blocks_max; off by default).
    Without it every decode maps ~8 MB afresh and unmaps it again: with a pool of decode threads that is a page-fault and mmap-lock
    storm inside one process -- measured here on 8 cores, 1080x1920 JPEGs: 4 threads 75 frames/s without the cache (51 ms per
    call against 12 ms alone), 260 frames/s with it.  Harmless where the hook does not exist."""
    global _pil_cache_blocks
    try:
        from PIL import Image
        want = max(int(n_images), 0) * 2                      # 16 MB blocks; an RGB 1080p image takes one, a 4K image three
        if want > _pil_cache_blocks and want > Image.core.get_blocks_max():
            Image.core.set_blocks_max(want)
            _pil_cache_blocks = want
    except Exception:
        pass


def read_npy_into(path, dst: np.ndarray) -> bool:
    """An .npy file's payload straight into dst (same dtype, shape and C order; e.g. pinned staging) with ONE read(2) -- no
    mapping of the file (2000 minor page faults per 1080p frame, on the process-wide mmap lock: 4 decode threads reached 400
    frames/s through np.load(mmap_mode="r") + memmove and 1650 through this), no interpreter lock while the bytes move.
    False when the file holds something else (the caller falls back to np.load)."""
    fmt = np.lib.format
    try:
        with open(str(path), "rb") as f:
            ver = fmt.read_magic(f)
            if ver == (1, 0):
                shape, fortran, dt = fmt.read_array_header_1_0(f)
            elif ver == (2, 0):
                shape, fortran, dt = fmt.read_array_header_2_0(f)
            else:
                return False
            if fortran or dt != dst.dtype or tuple(shape) != dst.shape or dt.hasobject or not dst.flags["C_CONTIGUOUS"]:
                return False
            got = f.readinto(memoryview(dst).cast("B"))
            return got == dst.nbytes
    except (OSError, ValueError):
        return False


def copy_bytes(dst: np.ndarray, src, nbytes: int):
    """memcpy into a C-contiguous array without the interpreter lock (ctypes releases it around foreign calls): the
    decode workers of FramePrefetcher copy into their pinned staging buffers side by side."""
    import ctypes
    if isinstance(src, np.ndarray):
        src = src.ctypes.data
    ctypes.memmove(dst.ctypes.data, src, int(nbytes))


def resize_bilinear(img: np.ndarray, width: int, height: int) -> np.ndarray:
    """cv2.resize(img, (width, height), interpolation=cv2.INTER_LINEAR) for a float32 single-channel image
    (D2R:465-467): source coordinate (x + 0.5) * scale - 0.5, weights in float32, border replicated."""
    src = np.asarray(img, dtype=np.float32)
    sh, sw = src.shape

    def taps(dn, sn):
        scale = np.float64(sn) / np.float64(dn)
        f = ((np.arange(dn, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        i0 = np.floor(f).astype(np.int64)
        w = (f - i0.astype(np.float32)).astype(np.float32)
        lo = i0 < 0
        i0[lo], w[lo] = 0, 0.0
        hi = i0 >= sn - 1
        i0[hi], w[hi] = sn - 1, 0.0
        return i0, np.minimum(i0 + 1, sn - 1), w

    x0, x1, wx = taps(width, sw)
    y0, y1, wy = taps(height, sh)
    rows = src[:, x0] * (np.float32(1.0) - wx)[None, :] + src[:, x1] * wx[None, :]
    out = rows[y0, :] * (np.float32(1.0) - wy)[:, None] + rows[y1, :] * wy[:, None]
    return out.astype(np.float32)


def load_data(rgb_folder, depth_folder, verbose: bool = True):
    """RGB images sorted by name (.png/.jpg/.jpeg) each with its depth map, resized to the RGB size if needed
    (D2R:439-477).  Returns (images_bgr, depths_f32, names)."""
    rgb_path, depth_path = Path(rgb_folder), Path(depth_folder)
    files = sorted(f for f in rgb_path.iterdir() if f.suffix.lower() in IMAGE_SUFFIXES)
    if verbose:
        print(f"Found {len(files)} RGB images")
    images, depths, names = [], [], []

    def one(f):                                        # decode of one pair (PIL's decoders and file reads run outside the interpreter lock)
        img = read_image_bgr(f)
        if img is None:
            return None, None, None
        dfile = DepthImageLoader.find_matching_depth(f.name, depth_path)
        if dfile is None:
            return img, None, None
        depth = DepthImageLoader.load_depth(dfile)
        if depth is not None and depth.shape[:2] != img.shape[:2]:
            depth = resize_bilinear(depth, img.shape[1], img.shape[0])
        return img, dfile, depth

    from concurrent.futures import ThreadPoolExecutor
    keep_pil_blocks(2 * min(16, max(1, usable_cpus())))
    with ThreadPoolExecutor(max_workers=min(16, max(1, usable_cpus()))) as pool:
        decoded = pool.map(one, files)                 # results in file order: the messages and lists are the serial loop's
        for f, (img, dfile, depth) in zip(files, decoded):
            if img is None:
                continue
            if dfile is None:
                if verbose:
                    print(f"  Warning: No depth found for {f.name}")
                continue
            if depth is None:
                continue
            images.append(img)
            depths.append(depth)
            names.append(f.name)
            if verbose:
                print(f"  Loaded: {f.name} with depth")
    if verbose:
        print(f"Loaded {len(images)} image-depth pairs")
    return images, depths, names


def usable_cpus() -> int:
    """CPUs this process may use: affinity mask capped by the cgroup CPU quota (containers show every core of the host)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


class FramePrefetcher:
    """Row f2: decode files on worker threads into pinned staging buffers and hand them to the device with asynchronous
    uploads, so decode, PCIe copy and GPU work overlap and host RAM holds `n_staging` frames instead of all of them
    (the reference keeps every decoded frame in two Python lists, D2R:434-437).

    for index, slot in FramePrefetcher(ctx, rgb_files, depth_files): ...   # frame `index` is (being) uploaded to `slot`
    Slots are assigned round-robin over ctx.n_slots; with n_slots >= number of frames every frame stays resident.
    """

    def __init__(self, ctx, rgb_files, depth_files, n_staging: int = 0, workers: int = 0, raw_u16: bool = True):
        from concurrent.futures import ThreadPoolExecutor
        assert len(rgb_files) == len(depth_files)
        self.ctx, self.rgb_files, self.depth_files = ctx, list(rgb_files), list(depth_files)
        self.raw_u16 = raw_u16
        h, w = ctx.height, ctx.width
        # Decode is the bottleneck of a file-fed run (JPEG ~6 ms, .npy ~2 ms, 16-bit PNG ~10 ms per 1080p frame and thread): one
        # worker per granted CPU but two.  Frames are handed on IN ORDER but decoded out of order, so a ring of workers + 2
        # staging buffers left the workers idle 40 % of the time behind the slowest frame (7.9 of 14 busy, 930 frames/s; a ring of
        # 2 x workers + 4: 13.5 busy, 1470 frames/s -- tools/bench_prefetch.py).  A staging buffer is page-locked memory of ONE kind
        # (f32 depth, 16-bit depth, colour), taken from the process-wide cache when ring position b is first handed to a worker
        # (by the iterating thread, whose current device is the context's): a data set of .npy depth never pays for the 16-bit
        # ring, and the locking of position b runs beside the decodes of the positions before it.
        workers = int(workers) if workers else min(16, max(2, usable_cpus() - 2))
        self.workers = max(1, int(workers))
        self.n_staging = max(2, int(n_staging) if n_staging else 2 * self.workers + 4)
        self._shape = {"f32": ((h, w), np.float32), "u16": ((h, w), np.uint16), "bgr": ((h, w, 3), np.uint8)}
        self._bufs = {k: [None] * self.n_staging for k in self._shape}
        keep_pil_blocks(2 * self.workers)                   # a colour and a depth image per worker
        self._pool = ThreadPoolExecutor(max_workers=self.workers)
        self.decode_s = 0.0

    def _ensure(self, i, b):
        """the staging buffers frame i may need at ring position b (iterating thread, before the decode is submitted)"""
        from .fusion import PinnedCache
        suffix = Path(self.depth_files[i]).suffix
        kinds = ["bgr"] if self.rgb_files[i] is not None else []
        if suffix == ".npy":
            kinds.append("f32")
        elif suffix == ".png" and self.raw_u16:
            kinds.append("u16")                             # (a PNG that is not 16-bit gray, or needs resizing, is rare: see _buf)
        else:
            kinds += ["f32", "u16"]
        for k in kinds:
            if self._bufs[k][b] is None:
                self._bufs[k][b] = PinnedCache.take(*self._shape[k])

    def _buf(self, kind, b) -> np.ndarray:
        """staging buffer b of one kind (only the worker that holds ring position b touches entry b)"""
        pa = self._bufs[kind][b]
        if pa is None:                                      # a file that is not what its suffix promised
            from .fusion import PinnedCache
            pa = self._bufs[kind][b] = PinnedCache.take(*self._shape[kind])
        return pa.array

    def _decode(self, i, b):
        """One frame into staging buffer b.  Everything bulky runs outside the interpreter lock: PIL's decoders release it,
        the copies into pinned memory are ctypes memmoves, an .npy depth map is read straight into the staging buffer."""
        import time
        t0 = time.perf_counter()
        h, w = self.ctx.height, self.ctx.width
        bgr = None
        if self.rgb_files[i] is not None and decode_bgr_into(self.rgb_files[i], self._buf("bgr", b)):
            bgr = self._buf("bgr", b)
        path = Path(self.depth_files[i])
        dst = None
        if path.suffix == ".npy" and read_npy_into(path, self._buf("f32", b)):
            dst = self._buf("f32", b)
        if dst is None and path.suffix == ".png" and self.raw_u16:
            from PIL import Image
            try:
                with Image.open(str(path)) as im:
                    if im.mode in ("I;16", "I;16L") and im.size == (w, h):
                        im.load()
                        rows = _pil_row_pointers(im, w, h, 2)
                        if rows is not None:
                            from . import _cabi as abi
                            dst = self._buf("u16", b)
                            abi.check(abi.load().tl3d_host_copy_rows(dst.ctypes.data, rows, h, 2 * w))
            except Exception:
                dst = None
        if dst is None:
            d = DepthImageLoader.load_depth(path, raw_u16=self.raw_u16)
            if d is None:
                raise IOError(f"cannot read depth {self.depth_files[i]}")
            if d.shape != (h, w):
                d = resize_bilinear(d.astype(np.float32) / (1000.0 if d.dtype == np.uint16 else 1.0), w, h)
            dst = self._buf("u16", b) if d.dtype == np.uint16 else self._buf("f32", b)
            d = np.ascontiguousarray(d, dtype=dst.dtype)
            copy_bytes(dst, d, dst.nbytes)
        self.decode_s += time.perf_counter() - t0
        return dst, bgr

    def __iter__(self):
        """Invariant: the frames that own a staging buffer (being decoded, or decoded and possibly still being copied)
        form a contiguous index range of length <= n_staging, so `index % n_staging` never collides."""
        from collections import deque
        n, S = len(self.depth_files), self.n_staging
        pending = {}                                  # frame index -> decode future
        in_flight = deque()                           # frame indices whose upload was enqueued but not yet waited for
        nxt = 0
        for i in range(n):
            while True:
                while nxt < n and len(pending) + len(in_flight) < S:
                    self._ensure(nxt, nxt % S)
                    pending[nxt] = self._pool.submit(self._decode, nxt, nxt % S)
                    nxt += 1
                if i in pending:
                    break
                self.ctx.slot_wait(in_flight.popleft() % self.ctx.n_slots)      # ring full of uploads: retire the oldest
            depth, bgr = pending.pop(i).result()
            slot = i % self.ctx.n_slots
            self.ctx.upload_async(slot, depth, bgr)
            in_flight.append(i)
            while len(in_flight) > 2:                  # an upload takes ~1 ms, a decode 10-30: the ring belongs to the decoders
                self.ctx.slot_wait(in_flight.popleft() % self.ctx.n_slots)
            yield i, slot
        while in_flight:
            self.ctx.slot_wait(in_flight.popleft() % self.ctx.n_slots)

    def close(self):
        from .fusion import PinnedCache
        self._pool.shutdown(wait=True)
        for group in self._bufs.values():
            for b, pa in enumerate(group):
                if pa is not None:
                    PinnedCache.give(pa)                     # stays page-locked for the next prefetcher (up to PinnedCache.limit_bytes)
                    group[b] = None


def save_depth_like_processor(depth_m: np.ndarray, out_dir, frame_id: str):
    """Write `<id>_depth.npy` and `<id>_depth.png` = (depth*1000).astype(uint16), the pair depth_processor.py:905-921
    produces (used by the synthetic-dataset writer and the plumbing test)."""
    from PIL import Image
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    np.save(str(out_dir / f"{frame_id}_depth.npy"), depth_m.astype(np.float32))
    mm = np.clip(depth_m * 1000.0, 0, 65535).astype(np.uint16)
    Image.fromarray(mm).save(str(out_dir / f"{frame_id}_depth.png"))


# ---------------------------------------------------------------------------------------------
# PLY
# ---------------------------------------------------------------------------------------------
_HEADER_TAIL = ["property uchar red", "property uchar green", "property uchar blue", "end_header"]


def write_ply_ascii(path, points, colors):
    """The reference's fallback writer byte for byte (D2R:689-701): `float` xyz printed with Python's str() of the
    numpy scalar, colours as integers."""
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\n")
        f.write(f"element vertex {len(points)}\n")
        f.write("property float x\nproperty float y\nproperty float z\n")
        f.write("\n".join(_HEADER_TAIL) + "\n")
        for p, c in zip(points, colors):
            f.write(f"{p[0]} {p[1]} {p[2]} {int(c[0])} {int(c[1])} {int(c[2])}\n")


def write_ply_binary(path, points, colors, double_xyz: bool = True):
    """binary_little_endian PLY, vertices only.  double_xyz=True is the layout Open3D's write_point_cloud emits for
    the reference (D2R:684-687: points are Vector3dVector, i.e. fp64) -- from Open3D's source as remembered, not
    verifiable in this image."""
    pts = np.asarray(points)
    col = np.asarray(colors, dtype=np.uint8)
    ft, fname = ("<f8", "double") if double_xyz else ("<f4", "float")
    rec = np.empty(len(pts), dtype=[("x", ft), ("y", ft), ("z", ft), ("r", "u1"), ("g", "u1"), ("b", "u1")])
    if len(pts):
        rec["x"], rec["y"], rec["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
        rec["r"], rec["g"], rec["b"] = col[:, 0], col[:, 1], col[:, 2]
    header = "\n".join(["ply", "format binary_little_endian 1.0", f"element vertex {len(pts)}",
                        f"property {fname} x", f"property {fname} y", f"property {fname} z"] + _HEADER_TAIL) + "\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(rec.tobytes())


def save_reconstruction(points, colors, output_path, ascii: bool = False) -> bool:
    """DepthToReconstructionPipeline.save_reconstruction (D2R:673-703): empty input prints `No points to save` and
    writes nothing; parent directories are created; prints `Saved to <path>`."""
    if len(points) == 0:
        print("No points to save")
        return False
    filepath = Path(output_path)
    filepath.parent.mkdir(parents=True, exist_ok=True)
    if ascii:
        write_ply_ascii(filepath, points, colors)
    else:
        write_ply_binary(filepath, points, colors)
    print(f"Saved to {filepath}")
    return True
