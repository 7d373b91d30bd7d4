"""world_size-2 gloo tests (CPU) of the image-parallel plumbing: weight broadcast + round-robin shard."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from inklayer_amd import dist as idist
    r, w, _ = idist.init_process_group("gloo")
    spec = {"a.weight": ((300, 7), torch.float32), "b.bias": ((5,), torch.float32), "c.idx": ((11,), torch.int32),
            "d.big": ((1000, 1000), torch.float32)}
    sd = None
    if r == 0:
        g = torch.Generator().manual_seed(0)
        sd = {k: (torch.randn(s, generator=g) if dt.is_floating_point else torch.arange(11, dtype=dt))
              for k, (s, dt) in spec.items()}
    old = idist.BUCKET_BYTES
    idist.BUCKET_BYTES = 2_000_000            # force several buckets
    got = idist.broadcast_state_dict(spec, sd, "cpu")
    idist.BUCKET_BYTES = old
    chk = {k: float(v.double().sum()) for k, v in got.items()}
    # every received tensor starts on a 32-byte boundary of its flat bucket (b.bias follows the odd-sized a.weight)
    assert all(v.data_ptr() % 32 == 0 for v in got.values()), {k: v.data_ptr() % 32 for k, v in got.items()}
    mx = idist.max_over_ranks(float(r + 1), "cpu")
    idist.barrier()
    q.put((r, chk, idist.shard_indices(13, r, w), mx, {k: tuple(v.shape) for k, v in got.items()}))
    torch.distributed.destroy_process_group()


def test_broadcast_and_shard_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in ps])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, c0, s0, m0, sh0), (_, c1, s1, m1, sh1) = res
    assert c0 == c1 and sh0 == sh1 and sh0["a.weight"] == (300, 7)
    assert sorted(s0 + s1) == list(range(13)) and not set(s0) & set(s1)
    assert m0 == m1 == 2.0


def test_single_process_is_passthrough():
    from inklayer_amd import dist as idist
    sd = {"x": torch.ones(3)}
    out = idist.broadcast_state_dict({"x": ((3,), torch.float32)}, sd, "cpu")
    assert torch.equal(out["x"], sd["x"]) and idist.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]
    assert idist.max_over_ranks(3.5, "cpu") == 3.5


_RANK_SCRIPT = """
import json, os, sys, torch
sys.path.insert(0, %r)
from inklayer_amd import dist as idist
rank, world, local = idist.init_process_group("gloo")
assert world == 2 and os.environ["MASTER_ADDR"] == "127.0.0.1"
if len(sys.argv) > 1 and sys.argv[1] == "fail" and rank == 1:
    sys.exit(7)
t = idist.max_over_ranks(float(rank + 1), "cpu")
idist.barrier()
print("noise from rank", rank)
if rank == 0:
    print(json.dumps({"n_gpus": world, "max": t, "shard": idist.shard_indices(6, rank, world)}))
torch.distributed.destroy_process_group()
"""


def test_launch_ranks_spawns_world2_and_returns_rank0_stdout():
    """`bench.py --gpus N` without an external launcher goes through dist.launch_ranks: N fresh processes with the
    torchrun environment on 127.0.0.1, rank 0's stdout (the JSON line) handed back, non-zero if any rank fails."""
    import json
    import sys
    from pathlib import Path
    from inklayer_amd import dist as idist
    root = str(Path(__file__).resolve().parent.parent)
    rc, out = idist.launch_ranks([sys.executable, "-c", _RANK_SCRIPT % root], 2, timeout_s=300)
    assert rc == 0, out
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and "rank 1" not in out
    assert json.loads(lines[0]) == {"n_gpus": 2, "max": 2.0, "shard": [0, 2, 4]}
    rc, _ = idist.launch_ranks([sys.executable, "-c", _RANK_SCRIPT % root, "fail"], 2, timeout_s=300)
    assert rc == 7


def test_bench_self_launch_path_is_taken_without_touching_the_gpu(monkeypatch):
    """bench.main() with --gpus 2 and no WORLD_SIZE must delegate to launch_ranks (re-running this very script per
    rank) BEFORE any torch.cuda call."""
    import sys
    import importlib
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root))
    bench = importlib.import_module("bench")
    from inklayer_amd import dist as idist
    seen = {}

    def fake_launch(cmd, world, timeout_s=None, extra_env=None):
        seen["cmd"], seen["world"] = cmd, world
        return 0, '{"n_gpus": 2}\n'

    import torch
    monkeypatch.setattr(idist, "launch_ranks", fake_launch)
    monkeypatch.setattr(torch.cuda, "set_device", lambda *a, **k: (_ for _ in ()).throw(AssertionError("GPU touched")))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"])
    import pytest
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and seen["world"] == 2
    assert seen["cmd"][0] == sys.executable and seen["cmd"][1].endswith("bench.py") and "--gpus" in seen["cmd"]


def test_engines_do_not_pin_broadcast_buckets():
    """receivers of broadcast_state_dict get VIEWS into <= 1 GiB flat buckets; an engine parameter must be an
    owned copy (ops.own_f32), or a few KB of biases keep the whole bucket alive."""
    from inklayer_amd import ops
    flat = torch.zeros(100_000)
    view = flat[40:52].view(3, 4)
    owned = ops.own_f32(view, "cpu")
    assert torch.equal(owned, view) and owned.untyped_storage().nbytes() == 48
    whole = torch.ones(3, 4)
    assert ops.own_f32(whole, "cpu").data_ptr() == whole.data_ptr()          # already owns its storage: no copy
    assert ops.own_f32(whole.double(), "cpu").dtype == torch.float32
