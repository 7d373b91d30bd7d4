"""Synthetic sketches: white background, 40 random black polylines / ellipses,
2-6 px wide, seeded with numpy's frozen MT19937 stream.  There is no dataset offline."""
import numpy as np


def synthetic_sketch(seed: int, h: int = 1024, w: int = 1024) -> np.ndarray:
    from PIL import Image, ImageDraw
    rs = np.random.RandomState(seed)
    im = Image.new("RGB", (w, h), (255, 255, 255))
    d = ImageDraw.Draw(im)
    for _ in range(40):
        x0, y0, x1, y1 = (int(rs.randint(0, w)), int(rs.randint(0, h)), int(rs.randint(0, w)), int(rs.randint(0, h)))
        wd = int(rs.randint(2, 7))
        if rs.rand() < 0.5:
            d.line([(x0, y0), (x1, y1), (int(rs.randint(0, w)), int(rs.randint(0, h)))], fill=(0, 0, 0), width=wd)
        else:
            d.ellipse([min(x0, x1), min(y0, y1), max(x0, x1) + 1, max(y0, y1) + 1], outline=(0, 0, 0), width=wd)
    return np.ascontiguousarray(np.asarray(im))
