"""Development aid: bench.py with an engine attribute flipped, for same-box A/B runs.
usage: bench_ab.py <graph_blocks 0|1> [bench.py flags]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from inklayer_amd import sam
sam.SamEngine.graph_blocks = sys.argv[1] == "1"
del sys.argv[1]
import bench
bench.main()
