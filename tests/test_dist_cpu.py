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
