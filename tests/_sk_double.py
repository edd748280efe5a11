"""CPU test double of selavi_amd.sk_utils.HipSkBackend (same method protocol, numpy fp64 arithmetic
restated from the oracle).  Lives under tests/ only: it exists to exercise the multi-rank control flow
of ``sinkhorn`` / ``optimize_L_sk_gpu`` over gloo on machines without a GPU."""
import numpy as np
import torch


class _Ev:
    def synchronize(self):
        pass


class NumpySkBackend:
    name = "numpy-test-double"

    def device_of(self, t):
        return t.device

    def workspace(self, K, grid, device):
        # layout: [counter, done, err, pad] + alpha[K] + s[K+1]
        return torch.zeros(4 + K + K + 1, dtype=torch.float64)

    def default_grid(self, N, K):
        return 1

    def s_view(self, ws, K, grid):
        return ws[4 + K: 4 + K + K + 1]

    def alpha_view(self, ws, K, grid):
        return ws[4: 4 + K]

    def pow_(self, P, power):
        P.pow_(power)

    def colsum(self, P, weight, ws, grid):
        return (P * weight[:, None]).sum(0) if weight is not None else P.sum(0)

    def begin(self, P, N_global, beta, ws, grid):
        K = P.shape[1]
        ws[0], ws[1], ws[2] = 0, 0, 1e6
        beta.fill_(1.0 / N_global)
        self._pend = (beta[None, :] @ P).reshape(-1)       # s0 = beta^T P
        self._errp = 0.0

    def pass_(self, P, N_global, beta, ws, grid):
        if ws[1] != 0:
            return
        K = P.shape[1]
        alpha = self.alpha_view(ws, K, grid)
        t = P @ alpha
        bn = (1.0 / N_global) / t
        self._errp = float((beta / bn - 1.0).abs().sum()) if int(ws[0]) % 10 == 0 else 0.0
        beta.copy_(bn)
        self._pend = (bn[None, :] @ P).reshape(-1)

    def local_reduce(self, K, ws, grid):
        if ws[1] != 0:
            return
        s = self.s_view(ws, K, grid)
        s[:K] = self._pend
        s[K] = self._errp

    def update(self, r, K, tol, max_iter, first, ws, grid):
        if ws[1] != 0:
            return
        s = self.s_view(ws, K, grid)
        cnt = int(ws[0])
        err = float(ws[2])
        done = False
        if not first:
            if cnt % 10 == 0:
                err = float(s[K])
            cnt += 1
            done = (not (err > tol)) or cnt >= max_iter
        if not done:
            self.alpha_view(ws, K, grid).copy_(r / s[:K])
        ws[0], ws[1], ws[2] = cnt, float(done), err

    def iterate(self, P, beta, r, tol, max_iter, n_iters, ws, grid):
        K = P.shape[1]
        for _ in range(n_iters):
            self.pass_(P, P.shape[0], beta, ws, grid)
            self.local_reduce(K, ws, grid)
            self.update(r, K, tol, max_iter, False, ws, grid)

    def status_async(self, ws, K, grid, host_buf):
        host_buf[0], host_buf[1], host_buf[2] = ws[0], ws[1], ws[2]
        return _Ev()

    def labels(self, P, beta, ws, grid):
        K = P.shape[1]
        alpha = self.alpha_view(ws, K, grid)
        V = (P * beta[:, None]) * alpha[None, :]
        L = torch.argmax(V, 1)
        x = (V[torch.arange(P.shape[0]), L] * (1.0 / alpha[L])) * (1.0 / beta)
        lg = torch.log(x)
        return L, torch.nansum(lg).reshape(1)

    def host_status_buffer(self):
        return torch.zeros(4, dtype=torch.float64)
