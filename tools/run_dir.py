"""BASELINE config 5: the whole runner (detector, segmentor, mask cleanup, sketch NMS, Depth-Anything-V2, refinement)
over a directory of sketches, image-parallel: rank r takes files i with i % world == r (no collectives; every rank
loads / generates its own weights because the runner's plugins are process singletons) and feeds them, --batch at a
time, through the batched hot path (inklayer_amd/batch_runner.py).

    python tools/run_dir.py --dir sketches/ --out_dir output/                 # one GPU
    python tools/run_dir.py --dir sketches/ --out_dir output/ --gpus 8        # starts 8 ranks itself (one per GPU)
    python -m torch.distributed.run --nproc-per-node 8 tools/run_dir.py ...   # or under a launcher
Same outputs as the reference's `main.py --dir` (which loops serially, main.py:27-32)."""
import argparse
import glob
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", required=True)
    ap.add_argument("--out_dir", default="./output")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--no_intermediate", action="store_true")
    ap.add_argument("--batch", type=int, default=8, help="files per pass of the batched hot path")
    args = ap.parse_args()
    from inklayer_amd import dist as idist
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        rc, out0 = idist.launch_ranks([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], args.gpus)
        sys.stdout.write(out0)
        sys.exit(rc)
    rank, world, local = idist.env_rank_world()
    import torch
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    files = sorted(glob.glob(os.path.join(args.dir, "*.png"))) + sorted(glob.glob(os.path.join(args.dir, "*.jpg")))
    mine = [files[i] for i in idist.shard_indices(len(files), rank, world)]
    from inklayer_amd import batch_runner
    t0 = time.perf_counter()
    stages = {}
    batch_runner.run_files(mine, args.out_dir, batch=args.batch, no_intermediate=args.no_intermediate, stage_s=stages)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"[rank {rank}/{world}] {len(mine)} of {len(files)} sketches in {dt:.2f} s "
          f"({len(mine) / dt if dt > 0 else 0:.2f} sketches/s incl. the first-call set-up); stages: "
          + ", ".join(f"{k} {v:.2f} s" for k, v in stages.items()), flush=True)


if __name__ == "__main__":
    main()
