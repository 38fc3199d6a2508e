"""CPU tests (gloo, world_size 2): the multi-GPU path's sharding and gather, with the oracle
standing in for the per-rank generator (utterances are independent, so a rank's shard must equal
the corresponding rows of the unsharded run)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_is_a_partition():
    from nv_wavenet_amd.sharding import shard_range
    for B in (1, 7, 16, 21, 64):
        for G in (1, 2, 3, 8):
            spans = [shard_range(B, G, r) for r in range(G)]
            assert spans[0][0] == 0 and sum(n for _, n in spans) == B
            for (s0, n0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + n0 == s1
            assert max(n for _, n in spans) - min(n for _, n in spans) <= 1


def test_feature_shards_give_the_conditioning_of_the_batch_shard():
    """Round 5: with the conditioning computed in the kernel a rank is handed its rows of the features instead of its slice of Lh:
    Wcond x_shard + bcond must be the shard of Wcond x + bcond."""
    import cases
    import condgen
    from nv_wavenet_amd.sharding import shard_features, shard_inputs, shard_range
    cc = condgen.COND_BY_NAME["cond_oddL_B19"]
    s = cases.BY_NAME[cc.case_name].shape
    m = condgen.make_cond_model(cc, s)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((s.B, cc.n_cond, s.N)).astype(np.float32)
    w = m["cond_w"][:, :, 0]
    lh = (np.einsum("oc,bct->bot", w, x) + m["cond_b"][None, :, None]).reshape(s.B, s.L, 2 * s.R, s.N).transpose(3, 1, 0, 2)   # [N][L][B][2R]
    sel = rng.random((s.N, s.B), dtype=np.float32)
    for world in (2, 3, 8):
        for rank in range(world):
            xs = shard_features(x, world, rank)
            start, n = shard_range(s.B, world, rank)
            assert xs.shape == (n, cc.n_cond, s.N) and np.array_equal(xs, x[start:start + n])
            lh_r, _ = shard_inputs(np.ascontiguousarray(lh), sel, world, rank)
            mine = (np.einsum("oc,bct->bot", w, xs) + m["cond_b"][None, :, None]).reshape(n, s.L, 2 * s.R, s.N).transpose(3, 1, 0, 2)
            assert np.array_equal(np.ascontiguousarray(mine), lh_r)
            xt = shard_features(torch.from_numpy(x), world, rank)
            assert xt.data_ptr() == torch.from_numpy(x)[start:start + n].data_ptr() if n else True      # (a view, no copy)


def _worker(rank, world, port, total_batch, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cases
        import util
        from oracle import oracle as O
        from nv_wavenet_amd.sharding import shard_inputs, shard_range, gather_samples
        case = cases.BY_NAME["R32S128A256_impl1"]
        s = case.shape
        t = util.gen_inputs(case)
        Lh, sel = t.Lh[:, :, :total_batch], t.sel[:, :total_batch]
        Lh_r, sel_r = shard_inputs(np.ascontiguousarray(Lh), np.ascontiguousarray(sel), world, rank)
        start, n = shard_range(total_batch, world, rank)
        assert Lh_r.shape == (s.N, s.L, n, 2 * s.R) and sel_r.shape == (s.N, n)
        o = O.Oracle(s.L, n, s.N, s.R, s.S, s.A, s.maxD)
        o.set_model(t)
        o.set_inputs(Lh_r, sel_r)
        y_local = torch.from_numpy(o.run(s.N))
        y_full, _ = gather_samples(y_local, total_batch)
        _, fin = gather_samples(y_local, total_batch, async_op=True)
        assert torch.equal(fin(), y_full)
        # the streaming form (run_chunks: one gather per finished chunk, ragged last chunk): same result
        from nv_wavenet_amd.sharding import ChunkGatherer
        g = ChunkGatherer(total_batch, s.N, y_local)
        for first in range(0, s.N, 3):
            g(None, first, min(3, s.N - first))
        assert torch.equal(g.finish(), y_full)
        if rank == 0:
            q.put(y_full.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total_batch", [8, 7])
def test_two_rank_shards_equal_unsharded_run(total_batch):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    import util
    from oracle import oracle as O
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total_batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    y = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    case = cases.BY_NAME["R32S128A256_impl1"]
    s = case.shape
    t = util.gen_inputs(case)
    o = O.Oracle(s.L, total_batch, s.N, s.R, s.S, s.A, s.maxD)
    o.set_model(t)
    o.set_inputs(np.ascontiguousarray(t.Lh[:, :, :total_batch]), np.ascontiguousarray(t.sel[:, :total_batch]))
    y_ref = o.run(s.N)
    assert y.shape == (total_batch, s.N) and np.array_equal(y, y_ref)
