"""Test helper: the quality-adjusted score tables of vg's QualAdjAlignmentScorer
(src/alignment_scorer.cpp:30-99 recover_log_base, :438-492 qual_adjusted_matrix, :494-513 qual_adjusted_bonuses),
restated in numpy so raw C-ABI tests can build a quality-adjusted engine context and so the C++ shim's own
tables (vg_amd/host/aligner.cpp) have an independent cross-check."""
import math

import numpy as np


def recover_log_base(matrix16, gc=0.5, tol=1e-12):
    f = [0.5 * (1 - gc), 0.5 * gc, 0.5 * gc, 0.5 * (1 - gc)]

    def partition(lam):
        return sum(f[i] * f[j] * math.exp(lam * matrix16[i * 4 + j]) for i in range(4) for j in range(4))
    lam = 1.0
    part = partition(lam)
    if part < 1.0:
        lower = lam
        while part <= 1.0:
            lower = lam; lam *= 2.0; part = partition(lam)
        upper = lam
    else:
        upper = lam
        while part >= 1.0:
            upper = lam; lam /= 2.0; part = partition(lam)
        lower = lam
    while upper / lower - 1.0 > tol:
        lam = 0.5 * (lower + upper)
        if partition(lam) < 1.0:
            lower = lam
        else:
            upper = lam
    return 0.5 * (lower + upper)


def c_round(x):
    return int(math.floor(abs(x) + 0.5)) * (1 if x >= 0 else -1)     # std::round: half away from zero


def qual_adj_tables(match=1, mismatch=4, bonus=5, gc=0.5, max_qual=255):
    m16 = [match if i % 5 == 0 else -mismatch for i in range(16)]
    log_base = recover_log_base([float(x) for x in m16], gc)
    f = [0.5 * (1 - gc), 0.5 * gc, 0.5 * gc, 0.5 * (1 - gc)]
    ap = [math.exp(log_base * m16[i * 4 + j]) * f[i] * f[j] for i in range(4) for j in range(4)]
    acp = [sum(ap[i * 4 + k] for k in range(4) if k != j) for i in range(4) for j in range(4)]
    lowest = math.ceil(-10.0 * math.log10(0.75))
    mat = np.zeros((max_qual + 1) * 25, dtype=np.int8)
    for q in range(max_qual + 1):
        err = 10.0 ** (-q / 10.0)
        for i in range(5):
            for j in range(5):
                if i == 4 or j == 4 or q < lowest:
                    s = 0
                else:
                    s = c_round(math.log(((1.0 - err) * ap[i * 4 + j] + (err / 3.0) * acp[i * 4 + j])
                                         / (f[i] * ((1.0 - err) * f[j] + (err / 3.0) * (1.0 - f[j])))) / log_base)
                mat[q * 25 + i * 5 + j] = s
    p_full = math.exp(log_base * bonus) / (1.0 + math.exp(log_base * bonus))
    bon = np.zeros(max_qual + 1, dtype=np.int8)
    for q in range(lowest + 1, max_qual + 1):
        err = 10.0 ** (-q / 10.0)
        bon[q] = c_round(math.log(((1.0 - err * 4.0 / 3.0) * p_full + (err * 4.0 / 3.0) * (1.0 - p_full)) / (1.0 - p_full)) / log_base)
    return mat, bon
