"""CPU, world_size 2, gloo: the multi-GPU data path (frame sharding by rank + ONE all-gather of the
per-rank unit embeddings + each rank matching its own block) gives the same result as one process.
The GPU stages are replaced by the oracle here; the sharding / gather / indexing code is the
product's (pipeline.shard_range and the all_gather_into_tensor layout)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, q_all_np, gal, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import match
    from scrfd_arcface_facerecognition_amd.pipeline import shard_range
    n_total = q_all_np.shape[0]
    lo, hi = shard_range(n_total, world, rank)
    n = hi - lo
    assert n * world == n_total
    q_local = torch.from_numpy(q_all_np[lo:hi].copy())
    q_all = torch.empty((world * n, q_all_np.shape[1]), dtype=q_local.dtype)
    dist.all_gather_into_tensor(q_all, q_local)                    # the single collective
    assert torch.equal(q_all, torch.from_numpy(q_all_np))            # every rank holds all embeddings, in frame order
    mine = q_all[rank * n:(rank + 1) * n].numpy().astype(np.float32)
    idx, score = match.match_batch(mine, gal, 0.4)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), idx=idx, score=score, lo=lo, hi=hi)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    from scrfd_arcface_facerecognition_amd.pipeline import shard_range
    for n, w in ((512, 8), (64, 2), (10, 3), (7, 8)):
        spans = [shard_range(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


@pytest.mark.timeout(120)
def test_two_rank_gather_and_match(tmp_path):
    from oracle import match
    rng = np.random.default_rng(0)
    world, per_rank = 2, 8
    gal = rng.standard_normal((50, 512)).astype(np.float32)
    emb = rng.standard_normal((world * per_rank, 512)).astype(np.float32)
    for i in range(0, len(emb), 2):
        emb[i] = gal[rng.integers(0, 50)] + 0.5 * rng.standard_normal(512).astype(np.float32)
    q = (emb / np.linalg.norm(emb, axis=1, keepdims=True)).astype(np.float16)       # what fid_l2_normalize_f16 emits
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, q, gal, str(tmp_path)), nprocs=world, join=True)
    ref_idx, ref_score = match.match_batch(q.astype(np.float32), gal, 0.4)
    got_idx = np.concatenate([np.load(tmp_path / f"r{r}.npz")["idx"] for r in range(world)])
    got_score = np.concatenate([np.load(tmp_path / f"r{r}.npz")["score"] for r in range(world)])
    assert np.array_equal(got_idx, ref_idx) and np.allclose(got_score, ref_score)
    assert (got_idx >= 0).sum() >= per_rank - 1
