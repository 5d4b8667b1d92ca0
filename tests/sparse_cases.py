"""Helpers that turn dense batches into shared-pattern sparse batches (test data only)."""
import numpy as np
import scipy.sparse as sp


def dense_batch_to_sparse(P, A, n, m, upper_only=False):
    """P (B, n*n), A (B, m*n) col-major flat -> (Pp, Pi, Px, Ap, Aj, Ax) with the union pattern."""
    B = P.shape[0]
    Pm = P.reshape(B, n, n).transpose(0, 2, 1)
    Am = A.reshape(B, n, m).transpose(0, 2, 1)
    Pmask = (Pm != 0).any(0)
    if upper_only:
        Pmask = np.triu(Pmask)
    Amask = (Am != 0).any(0)
    Pc = sp.csc_matrix(Pmask.astype(float)); Pc.sort_indices()
    Ac = sp.csr_matrix(Amask.astype(float)); Ac.sort_indices()
    pcols = np.repeat(np.arange(n), np.diff(Pc.indptr))
    arows = np.repeat(np.arange(m), np.diff(Ac.indptr))
    Px = Pm[:, Pc.indices, pcols]
    Ax = Am[:, arows, Ac.indices]
    return (Pc.indptr.astype(np.int32), Pc.indices.astype(np.int32), np.ascontiguousarray(Px),
            Ac.indptr.astype(np.int32), Ac.indices.astype(np.int32), np.ascontiguousarray(Ax))
