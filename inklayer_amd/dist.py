"""Image-parallel multi-GPU support: one process per GPU, ONE collective (weight broadcast).

The path shards by independent units (sketches): rank r takes images i with i % world == r
(SURVEY §8e).  The only exchange is at start-up: rank 0 owns the checkpoint (or the seeded random
weights) and broadcasts them as a few large flat buffers over RCCL/xGMI (torch.distributed backend
"nccl" == RCCL on ROCm; "gloo" in the CPU tests).  No per-batch collectives.
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import torch
import torch.distributed as dist

BUCKET_BYTES = 1 << 30   # 1 GiB flat buckets: few, large messages (xGMI links are per-peer bound)


def env_rank_world() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_process_group(backend: str | None = None) -> Tuple[int, int, int]:
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            # INK_DIST_BACKEND=gloo: rehearsal hook (several ranks sharing one GPU, where RCCL refuses duplicates)
            backend = os.environ.get("INK_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        if "INK_FORCE_DEVICE" in os.environ:        # rehearsal hook: every rank on the same card
            local = int(os.environ["INK_FORCE_DEVICE"])
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif "INK_FORCE_DEVICE" in os.environ:
        local = int(os.environ["INK_FORCE_DEVICE"])
    return rank, world, local


def launch_ranks(cmd: List[str], world: int, timeout_s: float | None = None, extra_env: Dict[str, str] | None = None):
    """Start `world` fresh child processes of `cmd` (one per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT set, rendezvous on 127.0.0.1 at a free port), wait for all of them, and return
    (worst exit code, rank 0's stdout).  Used by `bench.py --gpus N` when it is started WITHOUT a launcher; the
    parent must not have touched the GPU (children are new processes, nothing is exec'ed over a HIP-initialised
    one).  If a rank dies, the others are terminated so that nobody waits forever in a collective."""
    import socket
    import subprocess
    import tempfile
    import time as _time
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")               # a file, not a pipe: rank 0 can never block on its stdout
    for r in range(world):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL across processes on this driver)
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen(cmd, env=env, stdout=out0 if r == 0 else subprocess.DEVNULL))
    t0 = _time.time()
    rc = 0
    pending = set(range(world))
    try:
        while pending:
            for r in list(pending):
                code = procs[r].poll()
                if code is not None:
                    pending.discard(r)
                    if code != 0:
                        rc = rc or code
            if rc != 0:
                break
            if timeout_s is not None and _time.time() - t0 > timeout_s:
                rc = 124
                break
            _time.sleep(0.05)
    finally:
        for r in pending:
            procs[r].terminate()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    out0.seek(0)
    text = out0.read()
    out0.close()
    return rc, text


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Static round-robin: item i -> rank i % world."""
    return list(range(rank, n_items, world))


def broadcast_state_dict(spec: Dict[str, Tuple[Tuple[int, ...], torch.dtype]],
                         sd: Dict[str, torch.Tensor] | None, device, src: int = 0) -> Dict[str, torch.Tensor]:
    """Every rank knows `spec` (name -> (shape, dtype)); only `src` needs `sd`.  Tensors travel packed into
    <= BUCKET_BYTES flat buffers per dtype, one dist.broadcast each; receivers get views into the buffers."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        assert sd is not None
        return {k: sd[k].to(device) for k in spec}
    out: Dict[str, torch.Tensor] = {}
    by_dtype: Dict[torch.dtype, List[str]] = {}
    for name, (_, dt) in spec.items():
        by_dtype.setdefault(dt, []).append(name)
    for dt, names in by_dtype.items():
        esz = torch.empty((), dtype=dt).element_size()
        align = max(1, 32 // esz)                      # elements per 32 bytes
        bucket: List[str] = []
        nbytes = 0

        def flush():
            nonlocal bucket, nbytes
            if not bucket:
                return
            numels = [int(torch.Size(spec[n][0]).numel()) for n in bucket]
            # every tensor starts on a 32-byte boundary of the flat buffer: the kernels read parameters with
            # 16-byte vector loads and must see the same alignment on every rank as torch's allocator gives rank 0
            slots = [-(-ne // align) * align for ne in numels]
            flat = torch.zeros(sum(slots), dtype=dt, device=device)
            if rank == src:
                off = 0
                for n, ne, sl in zip(bucket, numels, slots):
                    flat[off:off + ne].copy_(sd[n].reshape(-1))
                    off += sl
            dist.broadcast(flat, src=src)
            off = 0
            for n, ne, sl in zip(bucket, numels, slots):
                out[n] = flat[off:off + ne].view(spec[n][0])
                off += sl
            bucket, nbytes = [], 0

        for n in names:
            sz = -(-int(torch.Size(spec[n][0]).numel()) // align) * align * esz
            if nbytes + sz > BUCKET_BYTES and bucket:
                flush()
            bucket.append(n)
            nbytes += sz
        flush()
    return out


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier() -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
