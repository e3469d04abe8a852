"""Particle-filter analysis step (BASELINE config C5): host logic on CPU, N>1 over gloo.

The exchange plan and the all-to-all run here on CPU tensors with a stand-in batch that
implements pack_members / resample with the semantics of pf.hip; the GPU tests
(test_gpu_pf.py) check the real kernels against the same oracle (oracle/pf_oracle.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pf_oracle as po
from sipnet_amd import dist as sd

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORDS = 32 + 250


class FakeBatch:
    """CPU stand-in with pf.hip's packing layout: cols[words][ncol]"""

    def __init__(self, cols):
        self.cols = cols.clone()
        self.ncol = cols.shape[1]

    def pack_members(self, c, with_params=False):
        return self.cols[:, c.long()].contiguous()

    def resample(self, src, recv=None, block_cols=(), with_params=False):
        pool = [self.cols]
        off = 0
        for n in block_cols:
            pool.append(recv[off:off + WORDS * n].reshape(WORDS, n))
            off += WORDS * n
        self.cols = torch.cat(pool, dim=1)[:, src.long()].contiguous()


def weights_case(n, kind, seed=3):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return np.zeros(n)
    if kind == "mild":
        return -0.5 * rng.normal(size=n) ** 2
    if kind == "degenerate":                  # a handful of particles carry everything
        lw = np.full(n, -200.0)
        lw[rng.choice(n, 5, replace=False)] = rng.normal(size=5)
        return lw
    if kind == "one":
        lw = np.full(n, -np.inf)
        lw[n // 3] = 0.0
        return lw
    raise KeyError(kind)


@pytest.mark.parametrize("kind", ["uniform", "mild", "degenerate", "one"])
def test_systematic_resampling_properties(kind):
    n = 4096
    lw = weights_case(n, kind)
    w = po.fixed_weights(lw)
    for u0 in (0.0, 0.37, 0.999999999):
        a = po.systematic_ancestors(w, u0)
        assert (np.diff(a) >= 0).all() and a.min() >= 0 and a.max() < n
        cnt = np.bincount(a, minlength=n)
        assert (cnt[w == 0] == 0).all()                     # zero-weight particles never survive
        expect = n * w / w.sum()
        assert np.abs(cnt - expect).max() < 1.0 + 1e-9      # floor/ceil of the expected count
    if kind == "uniform":
        assert (po.systematic_ancestors(w, 0.5) == np.arange(n)).all()
    if kind == "one":
        assert (po.systematic_ancestors(w, 0.5) == n // 3).all()


def test_systematic_resampling_known_answers_and_the_textbook_walk():
    """the analysis oracle has no reference to be pinned to (PecanProject/sipnet has no filter): hand-computed cases, and the
    textbook formulation -- ONE uniform draw, n equally spaced pointers walked through the cumulative weights (Douc & Cappe 2005)
    -- restated independently with exact integer arithmetic: particle j takes the first i with n cdf[i] > (j + u0) S"""
    from fractions import Fraction
    # S = 8, pointers (j + 0.5) * 2 = 1, 3, 5, 7 against cdf = 1, 4, 4, 8 (strictly greater: a pointer ON a boundary moves on)
    assert po.systematic_ancestors(np.array([1, 3, 0, 4]), 0.5).tolist() == [1, 1, 3, 3]
    assert po.systematic_ancestors(np.array([1, 3, 0, 4]), 0.0).tolist() == [0, 1, 3, 3]          # pointers 0, 2, 4 (ON the boundary 4: moves on), 6
    assert po.systematic_ancestors(np.array([0, 0, 5, 0]), 0.25).tolist() == [2, 2, 2, 2]
    assert po.systematic_ancestors(np.array([2, 2, 2, 2]), 0.75).tolist() == [0, 1, 2, 3]
    rng = np.random.default_rng(11)
    for n in (1, 2, 7, 64, 1000):
        for _ in range(6):
            w = rng.integers(0, 1 << 30, size=n).astype(np.int64)
            w[rng.random(n) < 0.3] = 0
            if w.sum() == 0:
                w[n // 2] = 1
            u0 = float(rng.choice([0.0, 0.5, rng.random()]))
            got = po.systematic_ancestors(w, u0)
            S, cdf = int(w.sum()), np.cumsum(w)
            i, want = 0, []
            for j in range(n):
                ptr = (Fraction(j) + Fraction(u0)) * S          # (u0 is a double: exactly representable as a fraction)
                ptr = min(ptr, Fraction(n) * (S - 1))           # the oracle's (and the kernels') cap: never past the last weight
                while n * int(cdf[i]) <= ptr:
                    i += 1
                want.append(i)
            # the float pointer of the oracle is the exact one rounded: they may part only where a pointer sits within
            # rounding of a boundary of the cumulative weights
            diff = np.flatnonzero(got != np.array(want))
            for j in diff:
                exact = float((Fraction(int(j)) + Fraction(u0)) * S / n)
                assert min(abs(float(cdf[got[j]]) - exact), abs(float(cdf[want[j]]) - exact)) <= 4e-16 * max(exact, 1.0) * 4, (n, j)
            assert len(diff) <= max(1, n // 100)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("kind", ["uniform", "mild", "degenerate", "one"])
def test_exchange_plan_reassembles_the_global_gather(world, kind):
    """simulate every rank in one process: pack for each destination, hand the blocks over,
    resample -> equals the global gather; each needed ancestor crosses a link once"""
    n = 64
    N = n * world
    rng = np.random.default_rng(5)
    cols = [torch.from_numpy(rng.normal(size=(WORDS, n))) for _ in range(world)]
    anc = po.systematic_ancestors(po.fixed_weights(weights_case(N, kind)), 0.41)
    want = po.resample_global([c.numpy() for c in cols], anc)
    anc_t = torch.from_numpy(anc)
    plans = [sd.pf_exchange_plan(anc_t, n, world, r) for r in range(world)]
    batches = [FakeBatch(c) for c in cols]
    packed = [[batches[r].pack_members(plans[r][0][d]) for d in range(world)] for r in range(world)]
    for r in range(world):
        send_cols, src, recv_counts = plans[r]
        assert len(send_cols[r]) == 0 and recv_counts[r] == 0
        for s in range(world):       # sender and receiver agree on every block size
            assert recv_counts[s] == len(plans[s][0][r])
        recv = torch.cat([packed[s][r].reshape(-1) for s in range(world)])
        batches[r].resample(src, recv, recv_counts)
        np.testing.assert_array_equal(batches[r].cols.numpy(), want[r])
        needed = np.unique(anc[r * n:(r + 1) * n])
        remote = needed[(needed < r * n) | (needed >= (r + 1) * n)]
        assert sum(recv_counts) == len(remote)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, kind, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, REPO)
    rng = np.random.default_rng(5)
    cols = [torch.from_numpy(rng.normal(size=(WORDS, n))) for _ in range(world)]
    # each rank only knows its own log-weights; the gather makes them global
    lw_all = weights_case(n * world, kind)
    mine = torch.from_numpy(lw_all[rank * n:(rank + 1) * n].copy())
    gathered = sd._gather0(mine, world, None).reshape(-1).numpy()
    anc = torch.from_numpy(po.systematic_ancestors(po.fixed_weights(gathered), 0.41))
    b = FakeBatch(cols[rank])
    info = sd.pf_resample(b, anc, rank, world)
    np.save(os.path.join(out_dir, f"cols{rank}.npy"), b.cols.numpy())
    np.save(os.path.join(out_dir, f"info{rank}.npy"), np.array([info["sent"], info["received"]]))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("kind", ["mild", "degenerate"])
def test_two_rank_gloo_all_to_all_matches_global_gather(kind, tmp_path):
    world, n = 2, 96
    mp.spawn(_worker, args=(world, _free_port(), n, kind, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(5)
    cols = [rng.normal(size=(WORDS, n)) for _ in range(world)]
    anc = po.systematic_ancestors(po.fixed_weights(weights_case(n * world, kind)), 0.41)
    want = po.resample_global(cols, anc)
    sent = recv = 0
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"cols{r}.npy"), want[r])
        s, v = np.load(tmp_path / f"info{r}.npy")
        sent, recv = sent + s, recv + v
    assert sent == recv
    if kind == "degenerate":
        assert 0 < sent <= 10        # five surviving particles, each crosses at most once per rank
