"""Lane-level numpy emulation of tonic_amd/csrc/mlp64.hip's fused grad kernel (test infra).

It mirrors the kernel's index formulas one to one — LDS weight images, the "S" / "F"
register layouts, the transposes through the per-wave scratch and the final fold into the
flat gradient image — with `v_mfma_f32_32x32x2_f32` modelled from the operand layouts in
/opt/skills/guides/cdna_hip_programming.md §3:

    A: lane l holds A[i = l & 31][k = l >> 5]        B: lane l holds B[k = l >> 5][j = l & 31]
    D: lane l, register r holds D[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31]

So a layout / index bug in the kernel design shows up on CPU (tests/test_mfma_layout.py)
before any GPU minute is spent.  Arithmetic is float64: this checks indices, not rounding.
"""
import numpy as np

TS = 36
LANES = np.arange(64)
S_OF, H_OF = LANES & 31, LANES >> 5


def feat(q, h):
    return 32 * (q >> 4) + (q & 3) + 8 * ((q & 15) >> 2) + 4 * h


def mfma32(a, b, c):
    """c: [64 lanes, 16 regs]; a, b: [64]."""
    out = c.copy()
    for lane in range(64):
        j, hh = lane & 31, lane >> 5
        for r in range(16):
            i = (r & 3) + 8 * (r >> 2) + 4 * hh
            out[lane, r] += a[i] * b[j] + a[i + 32] * b[j + 32]
    return out


class Wave:
    def __init__(self, params, O, A, actor, norm=None):
        self.O, self.A, self.actor = O, A, actor
        self.KS1 = 2 if O <= 4 else 9 if O <= 18 else 16
        w1, b1, w2, b2 = params[:4]
        KS1 = self.KS1
        self.W1S = np.zeros((2, KS1, 64))
        for t in range(2):
            for st in range(KS1):
                for l in range(64):
                    kh, i = l >> 5, l & 31
                    k = 2 * st + kh
                    self.W1S[t, st, l] = w1[32 * t + i, k] if k < O else 0.0
        self.W2S = np.zeros((2, 32, 64))
        self.W2B = np.zeros((2, 32, 64))
        for t in range(2):
            for st in range(32):
                for l in range(64):
                    kh, i = l >> 5, l & 31
                    f = feat(st, kh)
                    self.W2S[t, st, l] = w2[32 * t + i, f]
                    self.W2B[t, st, l] = w2[f, 32 * t + i]
        self.B1P = np.array([[b1[feat(q, h)] for q in range(32)] for h in range(2)])
        self.B2P = np.array([[b2[feat(q, h)] for q in range(32)] for h in range(2)])
        if actor:
            log_scale, w3, b3 = params[4].reshape(-1), params[5], params[6]
            sp = np.log1p(np.exp(log_scale))
            self.sigma = np.clip(sp + 1e-8, 1e-4, 1.0)
            self.b3 = b3
        else:
            w3, self.b3 = params[4], params[5]
        self.nout = w3.shape[0]
        self.W3P = np.array([[[w3[a, feat(q, h)] for q in range(32)] for h in range(2)]
                             for a in range(self.nout)])
        self.norm = norm
        # accumulators
        self.gW2 = np.zeros((2, 2, 64, 16))
        self.gW1 = np.zeros((2, 64, 16))
        self.gW3 = np.zeros((2, self.nout, 64))
        self.gb1 = np.zeros((2, 64))
        self.gb2 = np.zeros((2, 64))
        self.gb3 = np.zeros((self.nout, 64))
        self.gsig = np.zeros((self.nout, 64))
        self.stats = np.zeros((4, 64))

    # ---- building blocks (same names as the kernel)
    def dense_tanh(self, w_img, bias_img, x):          # x: [64 lanes, KS]
        acc = [np.zeros((64, 16)), np.zeros((64, 16))]
        for t in range(2):
            for lane in range(64):
                acc[t][lane] = bias_img[lane >> 5, 16 * t:16 * t + 16]
        for st in range(x.shape[1]):
            for t in range(2):
                acc[t] = mfma32(w_img[t, st], x[:, st], acc[t])
        return np.tanh(np.concatenate(acc, axis=1))    # [64, 32] S layout

    def head_linear(self, h2):
        z = np.zeros((64, self.nout))
        for a in range(self.nout):
            part = np.array([h2[l] @ self.W3P[a, l >> 5] for l in range(64)])
            z[:, a] = part + part[LANES ^ 32] + self.b3[a]
        return z

    def scatter_S(self, T, v):
        for lane in range(64):
            for q in range(32):
                T[feat(q, lane >> 5), lane & 31] = v[lane, q]

    def gather_F(self, T, t):
        out = np.zeros((64, 16))
        for lane in range(64):
            c, kh = lane & 31, lane >> 5
            out[lane] = T[32 * t + c, 16 * kh:16 * kh + 16]
        return out

    def tile(self, n0, n, data):
        O, A, KS1 = self.O, self.A, self.KS1
        ns = n0 + S_OF
        valid = ns < n
        x = np.zeros((64, KS1))
        for lane in range(64):
            for st in range(KS1):
                k = 2 * st + (lane >> 5)
                if valid[lane] and k < O:
                    v = data['observations'][ns[lane], k]
                    if not self.actor:
                        v = (v - self.norm[0][k]) / self.norm[1][k]
                    x[lane, st] = v
        h1 = self.dense_tanh(self.W1S, self.B1P, x)
        h2 = self.dense_tanh(self.W2S, self.B2P, h1)
        z = self.head_linear(h2)
        counted = valid & (H_OF == 0)
        dzl = np.zeros((64, self.nout))
        if self.actor:
            loc = np.tanh(z)
            act = np.where(valid[:, None], data['actions'][np.minimum(ns, n - 1)], loc)
            dif = act - loc
            var = self.sigma ** 2
            logp = (-(dif ** 2) / (2 * var) - np.log(self.sigma) - 0.5 * np.log(2 * np.pi)).sum(1)
            old = np.where(valid, data['log_probs'][np.minimum(ns, n - 1)], logp)
            adv = np.where(valid, data['advantages'][np.minimum(ns, n - 1)], 0.0)
            ratio = np.exp(logp - old)
            lo, hi = data['clip']
            dead = ((ratio > hi) & (adv > 0)) | ((ratio < lo) & (adv < 0))
            g = np.where(dead | ~valid, 0.0, -(adv * ratio))
            self.stats[0] += np.where(counted, -np.minimum(adv * ratio, adv * np.clip(ratio, lo, hi)), 0)
            self.stats[1] += np.where(counted, old - logp, 0)
            self.stats[2] += np.where(counted & ((ratio > hi) | (ratio < lo)), 1.0, 0)
            self.stats[3] += counted
            dloc = g[:, None] * dif / var
            dzl = dloc * (1 - loc ** 2)
            self.gsig += np.where(counted, (g[:, None] * (dif ** 2 / (var * self.sigma) - 1 / self.sigma)).T, 0)
            self.gb3 += np.where(counted, dzl.T, 0)
        else:
            ret = np.where(valid, data['returns'][np.minimum(ns, n - 1)], 0.0)
            err = np.where(valid, z[:, 0] - ret, 0.0)
            dzl[:, 0] = 2 * err
            self.stats[0] += np.where(counted, err ** 2, 0)
            self.stats[1] += np.where(counted, z[:, 0], 0)
            self.stats[3] += counted
            self.gb3[0] += np.where(counted, dzl[:, 0], 0)

        T = np.full((64, TS), np.nan)
        DO = np.full((32, 8), np.nan)
        # dW3
        self.scatter_S(T, h2)
        for lane in range(32):
            DO[lane, :self.nout] = dzl[lane]
        hF = [self.gather_F(T, 0), self.gather_F(T, 1)]
        for lane in range(64):
            kh = lane >> 5
            for m in range(16):
                for a in range(self.nout):
                    d = DO[16 * kh + m, a]
                    self.gW3[0, a, lane] += hF[0][lane, m] * d
                    self.gW3[1, a, lane] += hF[1][lane, m] * d
        # dz2
        dh2 = np.zeros((64, 32))
        for lane in range(64):
            for a in range(self.nout):
                dh2[lane] += dzl[lane, a] * self.W3P[a, lane >> 5]
        dz2 = dh2 * (1 - h2 ** 2)
        # dh1 (transposed MFMA with W2B) and dz1
        acc = [np.zeros((64, 16)), np.zeros((64, 16))]
        for st in range(32):
            for t in range(2):
                acc[t] = mfma32(self.W2B[t, st], dz2[:, st], acc[t])
        dz1 = np.concatenate(acc, axis=1) * (1 - h1 ** 2)
        # dW2
        self.scatter_S(T, dz2)
        aF = [self.gather_F(T, 0), self.gather_F(T, 1)]
        self.gb2 += np.stack([aF[0].sum(1), aF[1].sum(1)])
        self.scatter_S(T, h1)
        bF = [self.gather_F(T, 0), self.gather_F(T, 1)]
        for m in range(16):
            for ti in range(2):
                for tj in range(2):
                    self.gW2[ti, tj] = mfma32(aF[ti][:, m], bF[tj][:, m], self.gW2[ti, tj])
        # dW1
        self.scatter_S(T, dz1)
        aF = [self.gather_F(T, 0), self.gather_F(T, 1)]
        self.gb1 += np.stack([aF[0].sum(1), aF[1].sum(1)])
        for lane in range(64):
            for st in range(KS1):
                T[2 * st + (lane >> 5), lane & 31] = x[lane, st]
        xF = self.gather_F(T, 0)
        xF = np.where(((LANES & 31) < 2 * KS1)[:, None], xF, 0.0)
        for m in range(16):
            for ti in range(2):
                self.gW1[ti] = mfma32(aF[ti][:, m], xF[:, m], self.gW1[ti])

    def fold(self):
        """Accumulators -> flat gradient image in the reference parameter order."""
        O, A = self.O, self.A
        oW1, ob1 = 0, 64 * O
        oW2 = ob1 + 64
        ob2 = oW2 + 4096
        oTail = ob2 + 64
        oLs = oTail
        oW3 = oTail + A if self.actor else oTail
        ob3 = oW3 + (A * 64 if self.actor else 64)
        P = ob3 + (A if self.actor else 1)
        G = np.zeros(P + 8)
        for lane in range(64):
            s, h = lane & 31, lane >> 5
            for ti in range(2):
                for r in range(16):
                    row = 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * h
                    G[oW2 + row * 64 + s] += self.gW2[ti, 0, lane, r]
                    G[oW2 + row * 64 + 32 + s] += self.gW2[ti, 1, lane, r]
                    if s < O:
                        G[oW1 + row * O + s] += self.gW1[ti, lane, r]
                if h == 0:
                    G[ob1 + 32 * ti + s] += self.gb1[ti, lane] + self.gb1[ti, lane ^ 32]
                    G[ob2 + 32 * ti + s] += self.gb2[ti, lane] + self.gb2[ti, lane ^ 32]
                    for a in range(self.nout):
                        G[oW3 + a * 64 + 32 * ti + s] += self.gW3[ti, a, lane] + self.gW3[ti, a, lane ^ 32]
        for a in range(self.nout):
            G[ob3 + a] += self.gb3[a].sum()
            if self.actor:
                G[oLs + a] += self.gsig[a].sum()
        G[P + 0], G[P + 1], G[P + 2], G[P + 5] = (self.stats[i].sum() for i in range(4))
        return G, P


def emulate_grad(params, O, A, actor, data, n, norm=None, waves=3):
    """Runs `waves` emulated waves over the tiles (round robin) and folds them."""
    ws = [Wave(params, O, A, actor, norm) for _ in range(waves)]
    ntiles = (n + 31) // 32
    for tile in range(ntiles):
        ws[tile % waves].tile(tile * 32, n, data)
    total, P = None, None
    for w in ws:
        G, P = w.fold()
        total = G if total is None else total + G
    return total, P
