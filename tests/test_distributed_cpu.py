"""N>1 path on CPU: two gloo ranks shard a batch, all-gather their codes (int16 on the wire) and must reproduce the
single-process result exactly.  The per-rank 'encoder' here is the oracle (test infrastructure) -- what is under test
is esc.distributed: shard bounds, narrowing/widening, equal and ragged all-gathers, rank order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, ragged, out_dir, use_bench_plan=False):
    sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd"))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import json
    from conftest import load_golden, synth_state
    from esc import synth
    from esc.distributed import all_gather_codes, shard_bounds
    from oracle.esc_oracle import EscOracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    g = load_golden("tiny")
    orc = EscOracle(json.loads(str(g["config_json"])), synth_state("tiny"))
    pcm = np.stack([synth.noise_clip_int16(f"dist-{i}", 1280) for i in range(total)])
    x = torch.from_numpy(synth.pcm_to_float(pcm))
    if use_bench_plan:                   # bench.py --global-batch: the strong-scaling shard rule of BASELINE configs[3]
        import bench
        lo, n_local, plan_counts = bench.shard_plan(total, world, rank)
        hi = lo + n_local
        assert (lo, hi) == shard_bounds(total, world, rank) and sum(plan_counts) == total
    elif ragged:
        lo, hi = shard_bounds(total, world, rank)
    else:
        per = total // world
        lo, hi = rank * per, (rank + 1) * per
    local, _ = orc.encode(x[lo:hi], 3)
    counts = [shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world)] if ragged else None
    full = all_gather_codes(local, counts=counts)
    pending = all_gather_codes(local, counts=counts, async_op=True)         # the overlapped form: same bytes once wait() has joined
    assert torch.equal(pending.wait(), full) and pending.wait() is pending.wait()
    if rank == 0:
        ref, _ = orc.encode(x[: (total if ragged else world * (total // world))], 3)
        np.save(os.path.join(out_dir, f"ok_{int(ragged)}.npy"), np.array([int(torch.equal(full, ref)), full.shape[0], int(full.dtype == torch.int64)]))
    elif use_bench_plan:                 # every rank must hold the whole batch, not only rank 0
        ref, _ = orc.encode(x, 3)
        assert torch.equal(full, ref)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("ragged,total", [(False, 6), (True, 5)])
def test_two_rank_all_gather_matches_single_process(tmp_path, ragged, total):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, total, ragged, str(tmp_path)), nprocs=2, join=True)
    ok = np.load(tmp_path / f"ok_{int(ragged)}.npy")
    assert ok[0] == 1 and ok[1] == (total if ragged else 2 * (total // 2)) and ok[2] == 1


def test_shard_bounds_and_code_narrowing():
    sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd"))
    from esc.distributed import narrow_codes, shard_bounds, widen_codes
    for total in (1, 7, 36, 288):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    codes = torch.randint(0, 1024, (4, 6, 3, 150))
    assert narrow_codes(codes).dtype == torch.int16
    assert torch.equal(widen_codes(narrow_codes(codes)), codes)


@pytest.mark.parametrize("world,total", [(4, 288), (4, 290), (3, 36)])
def test_strong_scaling_shards_of_the_node_batch(tmp_path, world, total):
    """BASELINE configs[3] on CPU ranks: the fixed 288-clip batch (and an uneven 290 / a 3-way split of 36) sharded with bench.py's
    own plan, one all-gather of the codes, every rank ends up with the single-process codes of the whole batch in rank order."""
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, True, str(tmp_path), True), nprocs=world, join=True)
    ok = np.load(tmp_path / "ok_1.npy")
    assert ok[0] == 1 and ok[1] == total and ok[2] == 1


def test_bench_shard_plan_is_a_partition():
    sys.path.insert(0, ROOT)
    import bench
    for total in (288, 36, 37):
        for world in (1, 2, 3, 4, 5, 7, 8):
            plans = [bench.shard_plan(total, world, r) for r in range(world)]
            assert plans[0][0] == 0 and sum(p[1] for p in plans) == total
            assert all(a[0] + a[1] == b[0] for a, b in zip(plans, plans[1:]))
            assert all(p[2] == plans[0][2] for p in plans) and plans[0][2] == [p[1] for p in plans]
    # the strong-mode clip tags are global: every world size processes the same 288 clips (first 36 = the weak-mode rank-0 batch)
    tags36 = [f"bench-r0-{i}" for i in range(36)]
    assert [f"bench-r{g // bench.CLIPS_PER_GPU}-{g % bench.CLIPS_PER_GPU}" for g in range(36)] == tags36


def _grad_worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "efficient-speech-codec_amd"))
    from esc.distributed import all_reduce_gradients
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(1_000_003, generator=g)
    all_reduce_gradients(flat, bucket_mb=1.0)             # 4 buckets, the last one ragged
    ref = sum(torch.randn(1_000_003, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)) / world
    np.save(os.path.join(out_dir, f"g{rank}.npy"), np.array([float((flat - ref).abs().max())]))
    dist.barrier(); dist.destroy_process_group()


def test_flat_gradient_all_reduce_is_the_mean_over_ranks(tmp_path):
    """Data-parallel training path: the bucketed all-reduce of the flat gradient buffer equals the mean over ranks on every rank."""
    port = _free_port()
    mp.spawn(_grad_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    for r in range(3):
        assert float(np.load(tmp_path / f"g{r}.npy")[0]) < 1e-6
