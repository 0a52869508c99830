"""Global-descriptor retrieval on the device: the kNN + recall@k part of the reference's `Evaluator.evaluate`
(eval/evaluate.py:60-88; MinkLocGLEvaluator.evaluate :168-184), i.e. the step that follows the database build of
BASELINE configs[4].  Arithmetic in libegonn_hip (no torch fallback)."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from . import _lib


def knn(query: torch.Tensor, database: torch.Tensor, k: int, chunk: int = 4096):
    """(Q,D), (M,D) -> indices (Q,k) int32 ascending by L2 distance (ties: lower index), distances (Q,k)."""
    dev = _lib.require_gpu() if not query.is_cuda else query.device
    lib = _lib.load()
    q = query.to(device=dev, dtype=torch.float32).contiguous()
    db = database.to(device=dev, dtype=torch.float32).contiguous()
    assert q.dim() == 2 and db.dim() == 2 and q.shape[1] == db.shape[1]
    nq, m = q.shape[0], db.shape[0]
    idx = torch.empty((nq, k), dtype=torch.int32, device=dev)
    dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
    scratch = torch.empty(min(max(nq, 1), chunk) * m, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        for lo in range(0, nq, chunk):
            hi = min(nq, lo + chunk)
            _lib.check(lib.egonn_knn(q[lo:hi].data_ptr(), hi - lo, db.data_ptr(), m, q.shape[1], k,
                                     idx[lo:hi].data_ptr(), dist[lo:hi].data_ptr(), scratch.data_ptr(), scratch.numel(),
                                     _lib._stream()))
    return idx, dist


def recall_at_k(map_embeddings: torch.Tensor, query_embeddings: torch.Tensor, map_positions: torch.Tensor,
                query_positions: torch.Tensor, radius: Sequence[float], k: int = 20,
                query_indexes: Optional[Sequence[int]] = None) -> Dict:
    """`Evaluator.evaluate` from the embeddings on: {'recall': {r: [recall@1 .. recall@k]}} (eval/evaluate.py:66-88;
    `query_indexes` = the reference's random sample of queries, all queries when None)."""
    dev = _lib.require_gpu() if not map_embeddings.is_cuda else map_embeddings.device
    lib = _lib.load()
    qe = query_embeddings.to(dev)
    # the reference keeps positions in float64 (eval/evaluate.py:66-88); MulRan / KITTI poses are UTM-scale (~4e6 m),
    # where fp32 resolves 0.25-0.5 m — enough to flip a neighbour at the 5 m / 20 m radius.  A common origin is
    # subtracted in float64 first, the device then works on metre-scale offsets.
    mp64 = torch.as_tensor(map_positions).to(dtype=torch.float64, device="cpu")
    qp64 = torch.as_tensor(query_positions).to(dtype=torch.float64, device="cpu")
    origin = mp64.mean(dim=0, keepdim=True) if mp64.shape[0] else torch.zeros((1, qp64.shape[1]), dtype=torch.float64)
    qp = (qp64 - origin).to(device=dev, dtype=torch.float32)
    if query_indexes is not None:
        sel = torch.as_tensor(list(query_indexes), dtype=torch.long, device=dev)
        qe, qp = qe[sel], qp[sel]
    qp = qp.contiguous()
    mp = (mp64 - origin).to(device=dev, dtype=torch.float32).contiguous()
    idx, _ = knn(qe, map_embeddings.to(dev), k)
    rad = torch.tensor([float(r) for r in radius], dtype=torch.float32, device=dev)
    tp = torch.empty((len(radius), k), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.egonn_recall_counts(idx.data_ptr(), qp.data_ptr(), mp.data_ptr(), qe.shape[0], k, qp.shape[1],
                                           rad.data_ptr(), len(radius), tp.data_ptr(), _lib._stream()))
    n = max(int(qe.shape[0]), 1)
    tpl = tp.cpu().tolist()
    return {'recall': {r: [c / n for c in tpl[i]] for i, r in enumerate(radius)}, 'nn_index': idx}
