"""CPU restatement of the particle-filter analysis step (test infrastructure only).

PARITY UNPINNED against the reference: PecanProject/sipnet has no particle filter -- PEcAn
runs one process per particle and moves SIPNET_RESTART files between cycles
(docs/developer-guide/restart-checkpoint.md).  What is restated here is the textbook
algorithm the engine implements (systematic resampling, e.g. Douc & Cappe 2005, with the
integer-weight variant of sipnet_amd/csrc/pf.hip so that results are exact) and the
semantics of the exchange: new particle g of the global set = old particle ancestors[g].

What pins it instead (tests/test_pf.py): hand-computed cases, the defining properties (every count the floor or ceiling of
its expectation, zero weights never survive, sorted ancestors) and an independent restatement of the textbook walk in exact
integer arithmetic.

Only tests/ may import this module.
"""
import numpy as np


def log_weights(plane, obs, sigma, status=None):
    """plane[T][ncol] -> -0.5*((sum_t plane - obs)/sigma)^2, summed in step order in fp64"""
    acc = np.zeros(plane.shape[1], dtype=np.float64)
    for t in range(plane.shape[0]):
        acc += plane[t].astype(np.float64)
    z = (acc - obs) * (1.0 / sigma)
    lw = -0.5 * z * z
    if status is not None:
        lw = np.where(status != 0, -np.inf, lw)
    return lw


def fixed_weights(logw):
    m = np.max(logw)
    with np.errstate(invalid="ignore"):
        e = np.exp(logw - m)
    w = np.where(np.isfinite(logw) | (logw > -np.inf), np.rint(e * 1073741824.0), 0.0)
    w = np.where(logw > -np.inf, w, 0.0)
    return w.astype(np.int64)


def systematic_ancestors(w_fixed, u0):
    """ancestor[j] = first i with cdf[i] > min(((j + u0) * S) / n, S - 1)"""
    n = len(w_fixed)
    cdf = np.cumsum(w_fixed.astype(np.int64))
    S = float(cdf[-1])
    assert S > 0 and S < 2.0 ** 53
    p = np.minimum(((np.arange(n, dtype=np.float64) + u0) * S) / float(n), S - 1.0)
    return np.searchsorted(cdf.astype(np.float64), p, side="right").astype(np.int32)


def resample_global(columns_by_rank, ancestors):
    """columns_by_rank: list over ranks of arrays [rows][n_local] -> same shape list, where
    global column g of the result is global column ancestors[g] of the input"""
    allc = np.concatenate(columns_by_rank, axis=1)
    new = allc[:, ancestors]
    n = columns_by_rank[0].shape[1]
    return [new[:, r * n:(r + 1) * n] for r in range(len(columns_by_rank))]
