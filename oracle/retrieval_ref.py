"""TEST INFRASTRUCTURE ONLY (CPU oracle; see oracle/me_ops.py header).  numpy restatement of the retrieval part of
the reference's Evaluator.evaluate (eval/evaluate.py:66-88; identical lines in MinkLocGLEvaluator.evaluate
:168-184).  The reference's np.argsort is not stable; ties are broken by the lower map index here (kind='stable'),
which is what the HIP path defines.  Pinned on the reference's own lines: plain numpy, importable anywhere."""
import numpy as np


def knn(query_embeddings, map_embeddings, k):
    idx = np.empty((len(query_embeddings), k), dtype=np.int32)
    dist = np.empty((len(query_embeddings), k), dtype=np.float32)
    for i, q in enumerate(query_embeddings):
        embed_dist = np.linalg.norm(map_embeddings - q, axis=1)                 # eval/evaluate.py:81
        nn = np.argsort(embed_dist, kind="stable")[:k]                           # :82
        idx[i, :len(nn)], dist[i, :len(nn)] = nn, embed_dist[nn]
        idx[i, len(nn):], dist[i, len(nn):] = -1, np.inf
    return idx, dist


def recall(nn_ndx, query_positions, map_positions, radius, k):
    tp = {r: [0] * k for r in radius}
    for qi in range(len(nn_ndx)):
        valid = nn_ndx[qi][nn_ndx[qi] >= 0]
        delta = query_positions[qi] - map_positions[valid]                      # :85
        euclid_dist = np.full(k, np.inf)
        euclid_dist[:len(valid)] = np.linalg.norm(delta, axis=1)                # :86
        tp = {r: [tp[r][nn] + (1 if (euclid_dist[:nn + 1] <= r).any() else 0) for nn in range(k)] for r in radius}   # :88
    n = max(len(nn_ndx), 1)
    return {r: [tp[r][nn] / n for nn in range(k)] for r in radius}              # :91
