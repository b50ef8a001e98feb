"""CPU tile backend for the multi-GPU driver's gloo tests: the same interface as HipTileBackend, every tile
operation done by numpy/scipy and the oracle's kernel evaluation.  TEST INFRASTRUCTURE ONLY - it lets the
block-cyclic schedule (ownership, broadcast order, index bookkeeping) run at world size 2 and 4 without a GPU."""
import numpy as np
from scipy.linalg import solve_triangular

from oracle import mgpi_oracle as oracle


def _terms(terms):
    return [(int(kind), float(sigma), np.asarray(lengths, dtype=float)) for kind, sigma, lengths in terms]


class NumpyTileBackend(object):
    def __init__(self, torch):
        self.torch = torch

    def zeros(self, rows, cols=None):
        shape = (rows,) if cols is None else (rows, cols)
        return self.torch.zeros(shape, dtype=self.torch.float64)

    def empty(self, rows, cols=None):
        t = self.zeros(rows, cols)
        t.fill_(float('nan'))          # poison: an element that is read before being written shows up in the results
        return t

    def from_numpy(self, a):
        return self.torch.from_numpy(np.array(a, dtype=np.float64, copy=True))

    def to_numpy(self, t):
        return t.numpy().copy()

    def fence(self):
        pass

    # the look-ahead's second stream: on the CPU everything simply runs in program order
    def side_backend(self):
        return self

    def record(self):
        return None

    def on_side(self, after):
        import contextlib
        return contextlib.nullcontext()

    def join_side(self):
        pass

    def cov(self, x_rows, x_cols, terms, out, n_rows_valid, n_cols_valid):
        out.zero_()
        if n_rows_valid and n_cols_valid:
            out.numpy()[:n_rows_valid, :n_cols_valid] = oracle.x2cov_chunked(_terms(terms), x_rows[:n_rows_valid],
                                                                             x_cols[:n_cols_valid])

    def potrf(self, block):
        a = block.numpy()
        sym = np.tril(a) + np.tril(a, -1).T
        try:
            L = np.linalg.cholesky(sym)
        except np.linalg.LinAlgError:
            return 1, 0.0
        a[:] = L
        return 0, float(np.sum(np.log(np.diag(L))))

    def trsm_right(self, lkk, rows):
        r = rows.numpy()
        r[:] = solve_triangular(np.tril(lkk.numpy()), r.T, lower=True).T

    def gemm_nt(self, c, a, b):
        c.numpy()[:] -= a.numpy() @ b.numpy().T

    def gemm_nt_stair(self, c, a, b, P, p, Q, q, bi0, bj0, nb):
        cn, an, bn = c.numpy(), a.numpy(), b.numpy()
        for bj in range(cn.shape[1] // nb):
            J = q + Q * (bj0 + bj)
            for bi in range(cn.shape[0] // nb):
                I = p + P * (bi0 + bi)
                if I >= J:
                    cn[bi * nb:(bi + 1) * nb, bj * nb:(bj + 1) * nb] -= an[bi * nb:(bi + 1) * nb] @ bn[bj * nb:(bj + 1) * nb].T

    def gemm_nn(self, c, a, b):
        c.numpy()[:] -= a.numpy() @ b.numpy()

    def trtri_neg(self, a_local, r0, c0, row_step, col_step, w, batch, out):
        a, o = a_local.numpy(), out.numpy()
        o[:batch * w, :w] = 0.0
        for j in range(batch):
            blk = np.tril(a[r0 + j * row_step:r0 + j * row_step + w, c0 + j * col_step:c0 + j * col_step + w])
            o[j * w:(j + 1) * w, :w] = -solve_triangular(blk, np.eye(w), lower=True)

    def trsm_left(self, lkk, rhs):
        r = rhs.numpy()
        r[:] = solve_triangular(np.tril(lkk.numpy()), r, lower=True)

    def trsv(self, lkk, vec, trans):
        v = vec.numpy()
        v[:] = solve_triangular(np.tril(lkk.numpy()), v, lower=True, trans='T' if trans else 'N')

    def gemv(self, a, x, y, trans):
        y.numpy()[:] -= (a.numpy().T if trans else a.numpy()) @ x.numpy()

    def predict_mean(self, xs, x, terms, alpha):
        return oracle.x2cov_chunked(_terms(terms), xs, x) @ alpha

    def colsq(self, v):
        a = v.numpy()
        return (a * a).sum(axis=0)

    def prior_diag(self, xs, terms):
        return oracle.paired_cov(_terms(terms), xs, xs)
