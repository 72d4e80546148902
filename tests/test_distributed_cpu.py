"""CPU, world_size 2 over gloo: the replica-sharding logic (shard ranges, seeded global noise slicing,
final all-gather) reproduces the single-process batch.  No HIP compute is involved."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    try:
        from perceptor_amd import distributed as D
        from perceptor_amd.utils.synth import seeded_noise
        r, _, w = D.init("gloo")
        noise = seeded_noise((n_total, 3, 8, 8), 1234)
        local = D.shard(noise, r, w)
        local = local * 0.5 + 0.5                      # stand-in for the per-sample sampling chain
        full = D.gather_images(local, n_total)
        q.put((rank, full))
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:                              # reported to the parent, which decides whether it is the port race
        q.put((rank, f"ERROR {type(e).__name__}: {e}"))
        raise


@pytest.mark.parametrize("n_total", [4, 5])
def test_two_rank_sharding_matches_single_process(n_total):
    from perceptor_amd.utils.synth import seeded_noise
    ctx = mp.get_context("spawn")

    def attempt():
        """-> {rank: gathered batch}, or the string "port busy" when (and only when) a rank failed to bind the rendezvous port."""
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
        for p in procs:
            p.start()
        try:
            out = dict(q.get(timeout=180) for _ in range(2))
            for p in procs:
                p.join(timeout=120)
        finally:
            for p in procs:
                if p.is_alive():
                    p.kill()          # exact child processes started above
        errors = [v for v in out.values() if isinstance(v, str)]
        if errors:
            if any("address already in use" in e.lower() or "eaddrinuse" in e.lower() for e in errors):
                return "port busy"
            raise AssertionError(f"a gloo rank failed: {errors}")
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        return out

    # the port found by _free_port() can be taken by another process before the ranks bind it: one retry with a new port, for that error only
    out = attempt()
    if out == "port busy":
        out = attempt()
    assert isinstance(out, dict), out
    ref = seeded_noise((n_total, 3, 8, 8), 1234) * 0.5 + 0.5
    assert torch.equal(out[0], ref) and torch.equal(out[1], ref)


def test_shard_ranges_cover_and_balance():
    from perceptor_amd.distributed import shard_range
    for n in (1, 7, 8, 64):
        for w in (1, 2, 3, 8):
            rs = [shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            sizes = [hi - lo for lo, hi in rs]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` without a torch.distributed.run environment starts its own ranks (the driver's invocation):
    --rehearse runs only the process-group plumbing (gloo on this GPU-less box), rank 0 prints the JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["all_gather_ok"] is True
    # a rank whose WORLD_SIZE disagrees with --gpus is rejected instead of silently benchmarking another layout
    env["WORLD_SIZE"] = "3"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "does not match" in (r.stderr + r.stdout)
