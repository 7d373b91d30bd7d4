"""File output helpers of this build (no counterpart in the reference, which saves every PNG serially through PIL /
cv2): a minimal PNG writer whose compression step is `zlib.compress` - it releases the GIL, so the dozens of mask files
of a sketch are encoded on a small thread pool - for the three pixel formats the runner's tree uses: bool [H, W] ->
1-bit grayscale (what PIL writes for mode "1" and reads back as mode "1"), uint8 [H, W] -> 8-bit grayscale (mode "L"),
uint8 [H, W, 3] -> RGB.  Pixels are stored losslessly, like any PNG; only the compression level (1, cv2.imwrite's
default ballpark) differs from PIL's default."""
import struct
import zlib

import numpy as np


def _chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def png_bytes(a: np.ndarray, level: int = 1) -> bytes:
    a = np.asarray(a)
    h, w = a.shape[:2]
    if a.dtype == np.bool_ and a.ndim == 2:
        rows, depth, ctype = np.packbits(a, axis=1), 1, 0
    elif a.dtype == np.uint8 and a.ndim == 2:
        rows, depth, ctype = a, 8, 0
    elif a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] == 3:
        rows, depth, ctype = a.reshape(h, w * 3), 8, 2
    else:
        raise ValueError(f"png_bytes: unsupported array {a.dtype} {a.shape}")
    raw = np.empty((h, rows.shape[1] + 1), np.uint8)
    raw[:, 0] = 0                                             # filter type 0 (None) on every scanline
    raw[:, 1:] = rows
    return (b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
            + _chunk(b"IDAT", zlib.compress(raw.tobytes(), level)) + _chunk(b"IEND", b""))


def write_png(path, a, level: int = 1) -> None:
    data = png_bytes(a, level)
    with open(path, "wb") as fh:
        fh.write(data)


_POOL = None
_PENDING = []
DEFER = False       # set by InkLayer.runner.finish_sketch: save_all(wait=None) then returns before the files are written


def _pool():
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=8)
    return _POOL


def _write(job):
    im, path = job
    write_png(path, np.asarray(im() if callable(im) else im))


def save_all(jobs, wait=True) -> None:
    """(array | PIL image | callable returning one, path) pairs -> PNG files, encoded on up to 8 threads.
    wait=False: returns at once; `flush()` waits for everything submitted so far (the runner flushes before it returns,
    so the host-side encoding of one sketch's files overlaps with the GPU stages that follow).  wait=None: deferred only
    inside the runner (DEFER), synchronous when a plugin function is called on its own."""
    if wait is None:
        wait = not DEFER
    futs = [_pool().submit(_write, j) for j in jobs]
    if wait:
        for f in futs:
            f.result()
    else:
        _PENDING.extend(futs)


def flush() -> None:
    while _PENDING:
        _PENDING.pop().result()
