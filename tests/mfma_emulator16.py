"""Lane-level numpy emulation of tonic_amd/csrc/mlp64x16.hip (16x16x4 tiles; test infra).
Mirrors the kernel's index formulas one to one, with v_mfma_f32_16x16x4_f32 modelled as
    A: lane l holds A[i = l & 15][k = l >> 4]      B: lane l holds B[k = l >> 4][j = l & 15]
    D: lane l, register r holds D[row = 4 * (l >> 4) + r][col = l & 15]
(/opt/skills/guides/cdna_hip_programming.md §3).  float64: checks indices, not rounding."""
import numpy as np

TS16 = 24
LANES = np.arange(64)
S_OF, G_OF = LANES & 15, LANES >> 4


def feat16(q, g):
    return 16 * (q >> 2) + 4 * g + (q & 3)


def mfma16(a, b, c):
    """c: [64, 4]; a, b: [64]."""
    out = c.copy()
    for lane in range(64):
        j, gg = lane & 15, lane >> 4
        for r in range(4):
            i = 4 * gg + r
            out[lane, r] += sum(a[i + 16 * k] * b[j + 16 * k] for k in range(4))
    return out


class Wave16:
    def __init__(self, params, O, A, actor, norm=None):
        self.O, self.A, self.actor = O, A, actor
        self.KS1 = 1 if O <= 4 else 5 if O <= 20 else 8
        self.XT = 1 if O <= 4 else 2
        w1, b1, w2, b2 = params[:4]
        KS1 = self.KS1
        self.W1I = np.zeros((4, KS1, 64))
        for row in range(64):
            for k in range(O):
                T, i, st, gg = row >> 4, row & 15, k >> 2, k & 3
                self.W1I[T, st, gg * 16 + i] = w1[row, k]
        self.W2S = np.zeros((4, 4, 64, 4))
        self.W2B = np.zeros((4, 4, 64, 4))
        for row in range(64):
            for col in range(64):
                T, i = row >> 4, row & 15
                st, gg = ((col >> 4) << 2) | (col & 3), (col >> 2) & 3
                self.W2S[T, st >> 2, gg * 16 + i, st & 3] = w2[row, col]
                T, i = col >> 4, col & 15
                st, gg = ((row >> 4) << 2) | (row & 3), (row >> 2) & 3
                self.W2B[T, st >> 2, gg * 16 + i, st & 3] = w2[row, col]
        self.B1P = np.array([[b1[feat16(q, g)] for q in range(16)] for g in range(4)])
        self.B2P = np.array([[b2[feat16(q, g)] for q in range(16)] for g in range(4)])
        if actor:
            log_scale, w3, self.b3 = params[4].reshape(-1), params[5], params[6]
            self.sigma = np.clip(np.log1p(np.exp(log_scale)) + 1e-8, 1e-4, 1.0)
        else:
            w3, self.b3 = params[4], params[5]
        self.nout = w3.shape[0]
        self.W3P = np.array([[[w3[a, feat16(q, g)] for q in range(16)] for g in range(4)]
                             for a in range(self.nout)])
        self.norm = norm
        self.gW2 = np.zeros((4, 4, 64, 4))
        self.gW1 = np.zeros((4, self.XT, 64, 4))
        self.gW3 = np.zeros((4, 64, 4))
        self.gb1 = np.zeros((4, 64))
        self.gb2 = np.zeros((4, 64))
        self.gHead = np.zeros((64, 4))
        self.stats = np.zeros((4, 64))

    def chain64(self, wimg, vin, acc):
        for c in range(4):
            for e in range(4):
                for T in range(4):
                    acc[T] = mfma16(wimg[T, c, :, e], vin[:, 4 * c + e], acc[T])
        return acc

    def bias(self, bimg):
        return [np.array([bimg[l >> 4, 4 * T:4 * T + 4] for l in range(64)]) for T in range(4)]

    def scatter(self, T, v):
        for lane in range(64):
            for q in range(16):
                T[feat16(q, lane >> 4), lane & 15] = v[lane, q]

    def gather(self, T, tile):
        return np.array([T[16 * tile + (l & 15), 4 * (l >> 4):4 * (l >> 4) + 4] for l in range(64)])

    def tile(self, n0, n, data):
        O, A, KS1 = self.O, self.A, self.KS1
        ns = n0 + S_OF
        valid = ns < n
        idx = np.minimum(ns, n - 1)
        x = np.zeros((64, KS1))
        for lane in range(64):
            for st in range(KS1):
                k = 4 * st + (lane >> 4)
                if valid[lane] and k < O:
                    v = data['observations'][ns[lane], k]
                    if not self.actor:
                        v = (v - self.norm[0][k]) / self.norm[1][k]
                    x[lane, st] = v
        acc = self.bias(self.B1P)
        for st in range(KS1):
            for T in range(4):
                acc[T] = mfma16(self.W1I[T, st], x[:, st], acc[T])
        h1 = np.tanh(np.concatenate(acc, axis=1))
        h2 = np.tanh(np.concatenate(self.chain64(self.W2S, h1, self.bias(self.B2P)), axis=1))
        z = np.zeros((64, self.nout))
        for a in range(self.nout):
            part = np.array([h2[l] @ self.W3P[a, l >> 4] for l in range(64)])
            part = part + part[LANES ^ 16]
            part = part + part[LANES ^ 32]
            z[:, a] = part + self.b3[a]
        counted = valid & (G_OF == 0)
        dzl = np.zeros((64, self.nout))
        if self.actor:
            loc = np.tanh(z)
            act = np.where(valid[:, None], data['actions'][idx], loc)
            dif = act - loc
            var = self.sigma ** 2
            logp = (-(dif ** 2) / (2 * var) - np.log(self.sigma) - 0.5 * np.log(2 * np.pi)).sum(1)
            old = np.where(valid, data['log_probs'][idx], logp)
            adv = np.where(valid, data['advantages'][idx], 0.0)
            ratio = np.exp(logp - old)
            lo, hi = data['clip']
            dead = ((ratio > hi) & (adv > 0)) | ((ratio < lo) & (adv < 0))
            gl = np.where(dead | ~valid, 0.0, -(adv * ratio))
            self.stats[0] += np.where(counted, -np.minimum(adv * ratio, adv * np.clip(ratio, lo, hi)), 0)
            self.stats[1] += np.where(counted, old - logp, 0)
            self.stats[2] += np.where(counted & ((ratio > hi) | (ratio < lo)), 1.0, 0)
            self.stats[3] += counted
            dzl = gl[:, None] * dif / var * (1 - loc ** 2)
            dsg = gl[:, None] * (dif ** 2 / (var * self.sigma) - 1 / self.sigma)
        else:
            ret = np.where(valid, data['returns'][idx], 0.0)
            err = np.where(valid, z[:, 0] - ret, 0.0)
            dzl[:, 0] = 2 * err
            self.stats[0] += np.where(counted, err ** 2, 0)
            self.stats[1] += np.where(counted, z[:, 0], 0)
            self.stats[3] += counted
            dsg = np.zeros((64, self.nout))

        TA, TB = np.full((64, TS16), np.nan), np.full((64, TS16), np.nan)
        DO = np.zeros((16, 16))
        self.scatter(TA, h2)
        for lane in range(16):
            DO[lane, :self.nout] = dzl[lane]
            DO[lane, 8:8 + self.nout] = dsg[lane]
        dh2 = np.zeros((64, 16))
        for lane in range(64):
            for a in range(self.nout):
                dh2[lane] += dzl[lane, a] * self.W3P[a, lane >> 4]
        dz2 = dh2 * (1 - h2 ** 2)
        self.scatter(TB, dz2)
        dacc = self.chain64(self.W2B, dz2, [np.zeros((64, 4)) for _ in range(4)])
        aop = np.zeros((64, 4))
        for lane in range(64):
            i, gg = lane & 15, lane >> 4
            for e in range(4):
                aop[lane, e] = DO[4 * gg + e, i]
        for e in range(4):
            self.gHead = mfma16(aop[:, e], np.ones(64), self.gHead)
        for T in range(4):
            hF = self.gather(TA, T)
            for e in range(4):
                self.gW3[T] = mfma16(aop[:, e], hF[:, e], self.gW3[T])
        aF = [self.gather(TB, T) for T in range(4)]
        for T in range(4):
            self.gb2[T] += aF[T].sum(1)
        dz1 = np.concatenate(dacc, axis=1) * (1 - h1 ** 2)
        self.scatter(TA, h1)
        self.scatter(TB, dz1)
        for Tj in range(4):
            bF = self.gather(TA, Tj)
            for e in range(4):
                for Ti in range(4):
                    self.gW2[Ti, Tj] = mfma16(aF[Ti][:, e], bF[:, e], self.gW2[Ti, Tj])
        cF = [self.gather(TB, T) for T in range(4)]
        for T in range(4):
            self.gb1[T] += cF[T].sum(1)
        for lane in range(64):
            for st in range(KS1):
                TA[4 * st + (lane >> 4), lane & 15] = x[lane, st]
        for Tj in range(self.XT):
            xF = self.gather(TA, Tj)
            xF = np.where(((16 * Tj + (LANES & 15)) < 4 * KS1)[:, None], xF, 0.0)
            for e in range(4):
                for Ti in range(4):
                    self.gW1[Ti, Tj] = mfma16(cF[Ti][:, e], xF[:, e], self.gW1[Ti, Tj])

    def fold(self):
        O, A = self.O, self.A
        oW1, ob1 = 0, 64 * O
        oW2 = ob1 + 64
        ob2 = oW2 + 4096
        oTail = ob2 + 64
        oLs = oTail
        oW3 = oTail + A if self.actor else oTail
        ob3 = oW3 + (A * 64 if self.actor else 64)
        P = ob3 + self.nout
        G = np.zeros(P + 8)
        for lane in range(64):
            s, g = lane & 15, lane >> 4
            for Ti in range(4):
                for r in range(4):
                    row = 16 * Ti + 4 * g + r
                    for Tj in range(4):
                        G[oW2 + row * 64 + 16 * Tj + s] += self.gW2[Ti, Tj, lane, r]
                    for Tj in range(self.XT):
                        if 16 * Tj + s < O:
                            G[oW1 + row * O + 16 * Tj + s] += self.gW1[Ti, Tj, lane, r]
                    aa = 4 * g + r
                    if aa < self.nout:
                        G[oW3 + aa * 64 + 16 * Ti + s] += self.gW3[Ti, lane, r]
                if g == 0:
                    G[ob1 + 16 * Ti + s] += sum(self.gb1[Ti, s + 16 * k] for k in range(4))
                    G[ob2 + 16 * Ti + s] += sum(self.gb2[Ti, s + 16 * k] for k in range(4))
        for g in range(4):
            for r in range(4):
                row = 4 * g + r
                if row < 8 and row < self.nout:
                    G[ob3 + row] += self.gHead[16 * g, r]
                if self.actor and row >= 8 and row - 8 < self.nout:
                    G[oLs + row - 8] += self.gHead[16 * g, r]
        G[P + 0], G[P + 1], G[P + 2], G[P + 5] = (self.stats[k].sum() for k in range(4))
        return G, P


def emulate_grad16(params, O, A, actor, data, n, norm=None, waves=3):
    ws = [Wave16(params, O, A, actor, norm) for _ in range(waves)]
    for tile in range((n + 15) // 16):
        ws[tile % waves].tile(tile * 16, n, data)
    total, P = None, None
    for w in ws:
        G, P = w.fold()
        total = G if total is None else total + G
    return total, P
