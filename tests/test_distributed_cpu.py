"""CPU, world_size 2, gloo: the product's multi-GPU step `pipeline.run_step_distributed` itself runs at N > 1
(frame sharding by rank + ONE all-gather of the per-rank unit embeddings + the gallery match in its three scopes).
The device stages are replaced by a duck-typed pipe backed by the oracle (there is no GPU here); the sharding, the
collective calls, the gathered-matrix indexing and the sharded-gallery key exchange are the product's code."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _sortable(score_f32):
    u = score_f32.astype(np.float32).view(np.uint32).astype(np.uint64)
    neg = (u & 0x80000000) != 0
    return np.where(neg, (~u) & 0xFFFFFFFF, u | 0x80000000).astype(np.uint64)


class OraclePipe:
    """Same methods as FacePipeline (detect / embed / match / match_keys / match_merge, n_slots, idx, score); the
    arithmetic is the oracle's.  Device pointers are host pointers here."""

    def __init__(self, q_rank_f16, q_local, gal_rows):
        self.n_slots = q_rank_f16.shape[0]
        self._q_rank, self._q_local = q_rank_f16, q_local
        self.idx = torch.full((self.n_slots,), -7, dtype=torch.int32)
        self.score = torch.zeros((self.n_slots,), dtype=torch.float32)
        self.calls = []

    def detect(self, frames_dev, H, W):
        self.calls.append("detect")

    def embed(self, frames_dev, H, W):
        self.calls.append("embed")
        self._q_local.copy_(torch.from_numpy(self._q_rank))       # what fid_l2_normalize_f16 leaves in pipe.q

    @staticmethod
    def _rows(q, n):
        if isinstance(q, int):                                    # a raw address into the gathered matrix ("own" scope)
            return np.ctypeslib.as_array((C.c_uint16 * (n * 512)).from_address(q)).view(np.float16).reshape(n, 512).astype(np.float32)
        return q.numpy()[:n].astype(np.float32)

    def match(self, gallery, thresh, q=None, n=None, idx=None, score=None):
        from oracle import match
        i, s = match.match_batch(self._rows(q, n), gallery, thresh)
        (self.idx if idx is None else idx)[:n] = torch.from_numpy(i)
        (self.score if score is None else score)[:n] = torch.from_numpy(s)

    def match_keys(self, gallery, q, n, first_row, keys):
        e = self._rows(q, n)
        g = gallery / np.linalg.norm(gallery, axis=1, keepdims=True)
        s = (e / np.linalg.norm(e, axis=1, keepdims=True)) @ g.T
        j = s.argmax(axis=1)
        best = s[np.arange(n), j].astype(np.float32)
        k = (_sortable(best) << np.uint64(32)) | ((~(j.astype(np.uint64) + np.uint64(first_row))) & np.uint64(0xFFFFFFFF))
        keys.copy_(torch.from_numpy(k.view(np.int64)))

    def match_merge(self, keys_all, parts, n, G_total, thresh, idx, score):
        k = keys_all.numpy().view(np.uint64).reshape(parts, n).max(axis=0)
        u = (k >> np.uint64(32)).astype(np.uint32)
        u = np.where(u & 0x80000000, u & 0x7FFFFFFF, ~u).astype(np.uint32)
        s = u.view(np.float32)
        j = (~k & np.uint64(0xFFFFFFFF)).astype(np.int64)
        ok = (j < G_total) & (s > 0) & (s > thresh)
        idx[:n] = torch.from_numpy(np.where(ok, j, -1).astype(np.int32))
        score[:n] = torch.from_numpy(np.where(ok, s, 0).astype(np.float32))


F = 2          # face slots per frame in the ragged test


def _worker(rank, world, port, q_np, gal, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scrfd_arcface_facerecognition_amd.pipeline import run_step_distributed, shard_range
    N = q_np.shape[0]
    lo, hi = shard_range(N, world, rank)
    n = hi - lo
    assert n * world == N
    res = {}
    for scope in ("all", "own", "sharded"):
        q_local = torch.zeros((n, 512), dtype=torch.float16)
        q_all = torch.zeros((N, 512), dtype=torch.float16)
        idx_all = torch.full((N,), -9, dtype=torch.int32)
        score_all = torch.zeros((N,), dtype=torch.float32)
        glo, ghi = shard_range(len(gal), world, rank) if scope == "sharded" else (0, len(gal))
        pipe = OraclePipe(q_np[lo:hi], q_local, None)
        kw = {}
        if scope == "sharded":
            kw = dict(keys_local=torch.zeros((N,), dtype=torch.int64), keys_all=torch.zeros((world * N,), dtype=torch.int64),
                      gallery_first_row=glo, gallery_total=len(gal))
        run_step_distributed(pipe, None, 640, 640, gal[glo:ghi], 0.4, q_local, q_all, dist, idx_all=idx_all,
                             score_all=score_all, match_scope=scope, **kw)
        assert pipe.calls == ["detect", "embed"]
        assert torch.equal(q_all, torch.from_numpy(q_np))            # every rank holds all embeddings, in frame order
        # ... and with them the face counts of EVERY rank's frames (empty slots travel as zero rows)
        from scrfd_arcface_facerecognition_amd.pipeline import gathered_face_counts
        res[f"{scope}_counts"] = gathered_face_counts(q_all.numpy(), N // F, F)
        if scope == "own":
            res[f"{scope}_idx"], res[f"{scope}_score"] = pipe.idx.numpy().copy(), pipe.score.numpy().copy()
        else:
            res[f"{scope}_idx"], res[f"{scope}_score"] = idx_all.numpy().copy(), score_all.numpy().copy()
    # the pre-round-3 call shape (no idx_all / score_all) would write world*n results into the n-entry pipe.idx: refused at world > 1
    pipe = OraclePipe(q_np[lo:hi], torch.zeros((n, 512), dtype=torch.float16), None)
    try:
        run_step_distributed(pipe, None, 640, 640, gal, 0.4, torch.zeros((n, 512), dtype=torch.float16), torch.zeros((N, 512), dtype=torch.float16), dist)
        res["positional_raises"] = False
    except ValueError:
        res["positional_raises"] = True
    assert pipe.calls == []                                          # refused before any device work
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), lo=lo, hi=hi, **res)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    from scrfd_arcface_facerecognition_amd.pipeline import shard_range
    for n, w in ((512, 8), (64, 2), (10, 3), (7, 8), (100000, 8), (1000000, 8)):
        spans = [shard_range(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


@pytest.mark.timeout(180)
def test_two_rank_step_all_scopes(tmp_path):
    from oracle import match
    rng = np.random.default_rng(0)
    world, per_rank = 2, 8
    gal = rng.standard_normal((51, 512)).astype(np.float32)        # odd size: the two gallery shards differ in length
    gal[40] = gal[7]                                                # an exact duplicate in the OTHER shard: index 7 must win
    emb = rng.standard_normal((world * per_rank, 512)).astype(np.float32)
    for i in range(0, len(emb), 2):
        emb[i] = gal[rng.integers(0, 51)] + 0.5 * rng.standard_normal(512).astype(np.float32)
    emb[4] = gal[40]
    emb[12] = gal[50] + 0.1 * rng.standard_normal(512).astype(np.float32)   # best row = the last row of the last shard
    emb[4 * 2 + 1] = gal[33] + 0.1 * rng.standard_normal(512).astype(np.float32)   # slot 1 of frame 4: a match BEHIND a degenerate slot 0 (below)
    q = (emb / np.linalg.norm(emb, axis=1, keepdims=True)).astype(np.float16)       # what fid_l2_normalize_f16_slots emits ...
    # ... for ragged frames (F = 2 slots each): frame 1 has no face, frames 3 and 6 one face -- their empty slots are zero rows
    counts = np.full(world * per_rank // F, F, dtype=np.int32)
    counts[1], counts[3], counts[6] = 0, 1, 1
    for b, c in enumerate(counts):
        q[b * F + c:(b + 1) * F] = 0
        q[b * F + c:(b + 1) * F, 0] = -0.0                           # the empty-slot marker (csrc/match.hip l2norm_rows)
    # a DEGENERATE face inside a valid prefix (zero / NaN embedding -> all +0.0 row): slot 0 of the 2-face frame 4 and
    # the LAST valid slot of frame 5.  Both stay faces (reference main.py:132-134 visits every detected face): counts must not shrink
    # and slot 1 of frame 4 must keep its own match (round 3 counted non-zero rows and shifted it)
    q[4 * F + 0] = 0
    q[5 * F + 1] = 0
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, q, gal, str(tmp_path)), nprocs=world, join=True)
    with np.errstate(invalid="ignore", divide="ignore"):
        ref_idx, ref_score = match.match_batch(q.astype(np.float32), gal, 0.4)
    assert ref_idx[4] == 7 and ref_idx[12] == 50
    empty = np.array([f >= counts[b] for b in range(len(counts)) for f in range(F)])
    assert (ref_idx[empty] == -1).all() and (ref_score[empty] == 0).all()            # an empty slot is never a match
    assert ref_idx[4 * F] == -1 and ref_idx[5 * F + 1] == -1                          # ... nor is a degenerate face, but it is counted
    assert ref_idx[4 * F + 1] == 33                                                   # ... and the face behind it keeps its own slot
    outs = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    for r in range(world):
        assert bool(outs[r]["positional_raises"])
        for scope in ("all", "own", "sharded"):
            assert np.array_equal(outs[r][f"{scope}_counts"], counts), (r, scope)    # every rank recovers every frame's face count
    for r in range(world):                                          # "all" and "sharded": every rank holds the whole batch
        for scope in ("all", "sharded"):
            assert np.array_equal(outs[r][f"{scope}_idx"], ref_idx), (r, scope)
            assert np.allclose(outs[r][f"{scope}_score"], ref_score, atol=1e-6), (r, scope)
    own_idx = np.concatenate([outs[r]["own_idx"] for r in range(world)])           # "own": per-rank blocks, host concatenates
    own_score = np.concatenate([outs[r]["own_score"] for r in range(world)])
    assert np.array_equal(own_idx, ref_idx) and np.allclose(own_score, ref_score, atol=1e-6)
    assert (ref_idx >= 0).sum() >= per_rank - 2


class GroupedOraclePipe(OraclePipe):
    """the GroupedFacePipeline surface (collect / embed_collected, group): `steps_q` [steps, n, 512] are this rank's unit embeddings per step"""

    def __init__(self, steps_q, q_local, group):
        super().__init__(steps_q[0], q_local, None)
        self.group, self._steps_q, self._step, self.k = group, steps_q, 0, 0

    def collect(self, frames_dev, H, W):
        self.calls.append("collect")
        self.k += 1
        return self.k == self.group

    def embed_collected(self):
        steps = self.k
        if steps == 0:
            return 0
        self.calls.append(f"embed{steps}")
        for j in range(steps):
            self._q_local[j * self.n_slots:(j + 1) * self.n_slots].copy_(torch.from_numpy(self._steps_q[self._step + j]))
        self._step += steps
        self.k = 0
        return steps


def _worker_grouped(rank, world, port, q_steps, gal, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scrfd_arcface_facerecognition_amd.pipeline import run_step_distributed
    steps, _, n, _ = q_steps.shape                                   # [steps, world, n, 512]
    group = 2
    res = {}
    for scope in ("all", "sharded"):
        q_local = torch.zeros((group * n, 512), dtype=torch.float16)
        q_all = torch.zeros((world * group * n, 512), dtype=torch.float16)
        idx_all = torch.full((world * group * n,), -9, dtype=torch.int32)
        score_all = torch.zeros((world * group * n,), dtype=torch.float32)
        from scrfd_arcface_facerecognition_amd.pipeline import shard_range
        glo, ghi = shard_range(len(gal), world, rank) if scope == "sharded" else (0, len(gal))
        kw = {}
        if scope == "sharded":
            kw = dict(keys_local=torch.zeros((world * group * n,), dtype=torch.int64), keys_all=torch.zeros((world * world * group * n,), dtype=torch.int64),
                      gallery_first_row=glo, gallery_total=len(gal))
        pipe = GroupedOraclePipe(q_steps[:, rank], q_local, group)
        common = dict(idx_all=idx_all, score_all=score_all, match_scope=scope, **kw)
        done = [run_step_distributed(pipe, None, 640, 640, gal[glo:ghi], 0.4, q_local, q_all, dist, **common) for _ in range(2)]
        assert done == [False, True] and pipe.calls == ["collect", "collect", "embed2"]
        res[f"{scope}_full_idx"], res[f"{scope}_full_q"] = idx_all.numpy().copy(), q_all.numpy().copy()
        # a stream that ends inside a group: one more step, then flush = the same collective on the first n rows per rank
        idx_all.fill_(-9)
        assert run_step_distributed(pipe, None, 640, 640, gal[glo:ghi], 0.4, q_local, q_all, dist, **common) is False
        assert run_step_distributed(pipe, None, 640, 640, gal[glo:ghi], 0.4, q_local, q_all, dist, flush=True, **common) is True
        assert run_step_distributed(pipe, None, 640, 640, gal[glo:ghi], 0.4, q_local, q_all, dist, flush=True, **common) is False     # nothing left
        res[f"{scope}_part_idx"] = idx_all.numpy().copy()
    np.savez(os.path.join(out_dir, f"g{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_grouped_step(tmp_path):
    """run_step_distributed with a GroupedFacePipeline-shaped pipe (round 5): no collective until the group is full, then ONE all-gather of
    steps * n rows per rank in [rank][step][slot] order; flush of a partial group uses the first rows of the same buffers"""
    from oracle import match
    rng = np.random.default_rng(5)
    world, n, steps = 2, 4, 3
    gal = rng.standard_normal((37, 512)).astype(np.float32)
    emb = gal[rng.integers(0, 37, steps * world * n)] + 0.4 * rng.standard_normal((steps * world * n, 512)).astype(np.float32)
    q = (emb / np.linalg.norm(emb, axis=1, keepdims=True)).astype(np.float16).reshape(steps, world, n, 512)
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker_grouped, args=(world, port, q, gal, str(tmp_path)), nprocs=world, join=True)
    full = np.concatenate([q[s, r] for r in range(world) for s in range(2)])          # [rank][step][slot]
    part = np.concatenate([q[2, r] for r in range(world)])
    ref_full, _ = match.match_batch(full.astype(np.float32), gal, 0.4)
    ref_part, _ = match.match_batch(part.astype(np.float32), gal, 0.4)
    for r in range(world):
        o = np.load(tmp_path / f"g{r}.npz")
        for scope in ("all", "sharded"):
            assert np.array_equal(o[f"{scope}_full_q"], full), (r, scope)
            assert np.array_equal(o[f"{scope}_full_idx"], ref_full), (r, scope)
            assert np.array_equal(o[f"{scope}_part_idx"][:world * n], ref_part), (r, scope)
            assert (o[f"{scope}_part_idx"][world * n:] == -9).all(), (r, scope)       # rows behind the partial group are not written
