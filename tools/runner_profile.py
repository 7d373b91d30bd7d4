"""cProfile of the host side of the whole runner (config 5) over 8 synthetic 1024x1024 sketches: where the per-sketch
"tree + refinement" milliseconds go (development aid).  Usage: python tools/runner_profile.py [n_rows]"""
import cProfile
import os
import pstats
import shutil
import sys
import tempfile
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    import bench
    from PIL import Image
    from inklayer_amd import batch_runner, pipeline, synthetic
    os.environ["INKLAYER_RANDOM_WEIGHTS"] = "1"
    dev = torch.device("cuda:0")
    det, seg, _ = bench.build_engines(dev, 0, 1, 8)
    pipe = pipeline.InkLayerPipeline(det, seg)
    tmp = Path(tempfile.mkdtemp(prefix="ink_prof_"))
    try:
        (tmp / "in").mkdir()
        for i in range(8):
            Image.fromarray(synthetic.synthetic_sketch(100 + i)).save(tmp / "in" / f"s{i}.png")
        files = sorted(str(p) for p in (tmp / "in").glob("*.png"))
        batch_runner.run_files(files[:2], str(tmp / "warm"), batch=8, pipe=pipe, top_n=16)
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        batch_runner.run_files(files, str(tmp / "out"), batch=8, pipe=pipe, top_n=16)
        torch.cuda.synchronize()
        pr.disable()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 60)


if __name__ == "__main__":
    main()
