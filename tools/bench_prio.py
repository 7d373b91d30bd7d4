"""Development aid: bench.py with the pipeline's stream priorities set, for same-box A/B runs.
usage: bench_prio.py <det_priority> <seg_priority> [bench.py flags]   (-1 = high, 0 = default)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from inklayer_amd import pipeline
pipeline.InkLayerPipeline.det_priority = int(sys.argv[1])
pipeline.InkLayerPipeline.seg_priority = int(sys.argv[2])
del sys.argv[1:3]
import bench
bench.main()
