"""Lane-level numpy emulation of the SHIPPED form of tonic_amd/csrc/mlp64x16.hip (Lds16 CH = 3, grad_variant 4;
test infra): layer 1, the 64 x 64 products and the policy head's forward on 16x16x32 / 32x32x16 tiles, the head on
MFMA tiles with the loss distributed over the lane groups (policies with more than one action), one-output heads as
per-lane sums.  Mirrors the kernel's index formulas one to one — operand images, lane / register layouts, the
transposes through the LDS tiles, the fold into the flat gradient image — with
    v_mfma_f32_16x16x32_f16   A: lane l, slot e holds A[i = l & 15][k = 8 (l >> 4) + e]
                              B: lane l, slot e holds B[k = 8 (l >> 4) + e][j = l & 15]
                              D: lane l, register r holds D[row = 4 (l >> 4) + r][col = l & 15]
    v_mfma_f32_32x32x16_f16   A: lane l, slot e holds A[i = l & 31][k = 8 (l >> 5) + e]
                              B: lane l, slot e holds B[k = 8 (l >> 5) + e][j = l & 31]
                              D: lane l, register r holds D[row = 8 (r >> 2) + 4 (l >> 5) + (r & 3)][col = l & 31]
(/opt/skills/guides/cdna_hip_programming.md §3).  float64 throughout and one exact product where the kernel sums
three fp16 MFMAs: this checks INDICES; the arithmetic of the split is tests/test_fp16x2_arithmetic.py, the power-of-two
units cancel exactly."""
import numpy as np

from mfma_emulator16 import LANES, S_OF, G_OF, feat16, mfma16

TS = 20


def mfma16x32(a, b, c):
    """c: [64, 4]; a, b: [64, 8]."""
    A, B = np.zeros((16, 32)), np.zeros((32, 16))
    for lane in range(64):
        A[lane & 15, 8 * (lane >> 4):8 * (lane >> 4) + 8] = a[lane]
        B[8 * (lane >> 4):8 * (lane >> 4) + 8, lane & 15] = b[lane]
    D = A @ B
    out = c.copy()
    for lane in range(64):
        out[lane] += D[4 * (lane >> 4):4 * (lane >> 4) + 4, lane & 15]
    return out


def mfma32x16(a, b, c):
    """c: [64, 16]; a, b: [64, 8]."""
    A, B = np.zeros((32, 16)), np.zeros((16, 32))
    for lane in range(64):
        A[lane & 31, 8 * (lane >> 5):8 * (lane >> 5) + 8] = a[lane]
        B[8 * (lane >> 5):8 * (lane >> 5) + 8, lane & 31] = b[lane]
    D = A @ B
    out = c.copy()
    for lane in range(64):
        for r in range(16):
            out[lane, r] += D[8 * (r >> 2) + 4 * (lane >> 5) + (r & 3), lane & 31]
    return out


def row_sum(v):
    """v + the other 15 lanes of the DPP row, in every lane."""
    return np.array([v[16 * (l >> 4):16 * (l >> 4) + 16].sum() for l in range(64)])


def sum_groups(v):
    v = v + v[LANES ^ 16]
    return v + v[LANES ^ 32]


class WaveF16:
    def __init__(self, params, O, A, actor, norm=None):
        self.O, self.A, self.actor, self.norm = O, A, actor, norm
        self.KS1, self.XT, self.XR = ((1, 0, 4) if O <= 4 else (4, 1, 0) if O <= 16 else (5, 1, 1) if O == 17
                                      else (5, 1, 4) if O <= 20 else (8, 2, 0))
        self.XE = 4 if self.KS1 == 1 else 8
        w1, b1, w2, b2 = params[:4]
        # layer 1: [4 T][64 lanes][8]: lane (i, g), slot e = W1[16 T + i][XE g + e]
        self.W1I = np.zeros((4, 64, 8))
        for row in range(64):
            for k in range(O):
                T, i, gg, e = row >> 4, row & 15, k // self.XE, k % self.XE
                self.W1I[T, gg * 16 + i, e] = w1[row, k]
        # [2 m][4 T][64 lanes][8]: the K = 32 block m contracts the features feat16(8 m + e, g)
        self.W2S, self.W2B = np.zeros((2, 4, 64, 8)), np.zeros((2, 4, 64, 8))
        for row in range(64):
            for col in range(64):
                T, i = row >> 4, row & 15
                q, gg = ((col >> 4) << 2) | (col & 3), (col >> 2) & 3
                self.W2S[q >> 3, T, gg * 16 + i, q & 7] = w2[row, col]
                T, i = col >> 4, col & 15
                q, gg = ((row >> 4) << 2) | (row & 3), (row >> 2) & 3
                self.W2B[q >> 3, T, gg * 16 + i, q & 7] = w2[row, col]
        self.B1P = np.array([[b1[feat16(q, g)] for q in range(16)] for g in range(4)])
        self.B2P = np.array([[b2[feat16(q, g)] for q in range(16)] for g in range(4)])
        if actor:
            log_scale, w3, self.b3 = params[4].reshape(-1), params[5], params[6]
            self.sigma = np.clip(np.log1p(np.exp(log_scale)) + 1e-8, 1e-4, 1.0)
        else:
            w3, self.b3 = params[4], params[5]
        self.nout = w3.shape[0]
        AP = 1 if self.nout == 1 else 6 if self.nout <= 6 else 8
        self.HM = actor and AP > 1                       # Lds16::HM
        self.NS = (AP + 3) // 4 if self.HM else self.nout
        self.W3P = np.array([[[w3[a, feat16(q, g)] for q in range(16)] for g in range(4)]
                             for a in range(self.nout)])
        if self.HM:
            self.W3F = np.zeros((2, 64, 8))              # row 4 g' + r of the product = action g' + 4 r
            self.W3B = np.zeros((self.NS, 64, 4))        # step c: k = lane group = action g + 4 c
            for lane in range(64):
                i, gg = lane & 15, lane >> 4
                aa = (i >> 2) + 4 * (i & 3)
                for m in range(2):
                    for e in range(8):
                        if (i & 3) < self.NS and aa < self.nout:
                            self.W3F[m, lane, e] = w3[aa, feat16(8 * m + e, gg)]
                for c in range(self.NS):
                    if gg + 4 * c < self.nout:
                        for T in range(4):
                            self.W3B[c, lane, T] = w3[gg + 4 * c, 16 * T + i]
        self.gW2w = np.zeros((2, 2, 64, 16))
        self.gb2w = np.zeros((2, 64))
        self.gW1 = np.zeros((4, max(self.XT, 1), 64, 4))
        self.gW1r = np.zeros((4, max(self.XR, 1), 64))
        self.gb1 = np.zeros((4, 64))
        self.gW3 = np.zeros((4, 64, 4))
        self.gW3s = np.zeros((64, 16))                   # one-output heads: per-lane sums
        self.hb, self.hsg = np.zeros((64, 2)), np.zeros((64, 2))
        self.stats = np.zeros((4, 64))

    def bias(self, bimg):
        return [np.array([bimg[l >> 4, 4 * T:4 * T + 4] for l in range(64)]) for T in range(4)]

    def chain(self, wimg, vin, acc):
        for m in range(2):
            for T in range(4):
                acc[T] = mfma16x32(wimg[m, T], vin[:, 8 * m:8 * m + 8], acc[T])
        return acc

    @staticmethod
    def scatter(T, v):
        for lane in range(64):
            for q in range(16):
                T[feat16(q, lane >> 4), lane & 15] = v[lane, q]

    @staticmethod
    def gather(T, tile):
        return np.array([T[16 * tile + (l & 15), 4 * (l >> 4):4 * (l >> 4) + 4] for l in range(64)])

    @staticmethod
    def gather32(T, tile):
        return np.array([T[32 * tile + (l & 31), 8 * (l >> 5):8 * (l >> 5) + 8] for l in range(64)])

    def tile(self, n0, n, data):
        O, XE, NS = self.O, self.XE, self.NS
        ns = n0 + S_OF
        valid = ns < n
        idx = np.minimum(ns, n - 1)
        x = np.zeros((64, 8))
        for lane in range(64):
            for e in range(XE):
                k = XE * (lane >> 4) + e
                if valid[lane] and k < O:
                    v = data['observations'][ns[lane], k]
                    if not self.actor:
                        v = (v - self.norm[0][k]) / self.norm[1][k]
                    x[lane, e] = v
        z1 = self.bias(self.B1P)
        for T in range(4):
            z1[T] = mfma16x32(self.W1I[T], x, z1[T])
        h1 = np.tanh(np.concatenate(z1, axis=1))
        h2 = np.tanh(np.concatenate(self.chain(self.W2S, h1, self.bias(self.B2P)), axis=1))
        counted = valid & (G_OF == 0)
        dzl = np.zeros((64, NS))
        if self.HM:
            zacc = np.zeros((64, 4))
            for m in range(2):
                zacc = mfma16x32(self.W3F[m], h2[:, 8 * m:8 * m + 8], zacc)
            action = np.array([[(l >> 4) + 4 * r for r in range(NS)] for l in range(64)])
            live = action < self.A
            clamped = np.minimum(action, self.A - 1)
            z = zacc[:, :NS] + np.where(live, self.b3[clamped], 0.0)
            loc = np.tanh(z)
            act = np.where(valid[:, None], data['actions'][idx[:, None], clamped], loc)
            dif = act - loc
            sigma = self.sigma[clamped]
            var = sigma ** 2
            term = -(dif ** 2) / (2 * var) - np.log(sigma) - 0.5 * np.log(2 * np.pi)
            logp = sum_groups(np.where(live, term, 0.0).sum(1))
        elif self.actor:
            z = np.zeros((64, 1))
            part = np.array([h2[l] @ self.W3P[0, l >> 4] for l in range(64)])
            z[:, 0] = sum_groups(part) + self.b3[0]
            loc = np.tanh(z)
            act = np.where(valid[:, None], data['actions'][idx], loc)
            dif = act - loc
            sigma, live = self.sigma[None, :], np.ones((64, 1), bool)
            var = sigma ** 2
            logp = (-(dif ** 2) / (2 * var) - np.log(sigma) - 0.5 * np.log(2 * np.pi)).sum(1)
        if self.actor:
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
            dzl = np.where(live, gl[:, None] * dif / var * (1 - loc ** 2), 0.0)
            dsg = np.where(live, gl[:, None] * (dif ** 2 / (var * sigma) - 1 / sigma), 0.0)
        else:
            part = np.array([h2[l] @ self.W3P[0, l >> 4] for l in range(64)])
            z = sum_groups(part) + self.b3[0]
            ret = np.where(valid, data['returns'][idx], 0.0)
            err = np.where(valid, z - ret, 0.0)
            dzl = (2 * err)[:, None]
            dsg = np.zeros((64, 1))
            self.stats[0] += np.where(counted, err ** 2, 0)
            self.stats[1] += np.where(counted, z, 0)
            self.stats[3] += counted

        TA, TB = np.full((64, TS), np.nan), np.full((64, TS), np.nan)
        if self.HM:
            DOT = np.zeros((16, 16))                     # dO^T: row = action (8 .. 15 stay zero), 16 samples
            for lane in range(64):
                for r in range(NS):
                    DOT[(lane >> 4) + 4 * r, lane & 15] = dzl[lane, r]
            self.hb[:, :NS] += dzl
            self.hsg[:, :NS] += dsg
            self.scatter(TA, h2)
            hacc = [np.zeros((64, 4)) for _ in range(4)]
            for c in range(NS):
                for T in range(4):
                    hacc[T] = mfma16(self.W3B[c, :, T], dzl[:, c], hacc[T])
            dz2 = np.concatenate(hacc, axis=1) * (1 - h2 ** 2)
        else:                                            # one output: per-lane sums, no h2^T / dO tiles
            self.hb[:, 0] += dzl[:, 0]
            self.hsg[:, 0] += dsg[:, 0]
            self.gW3s += dzl[:, :1] * h2
            dh2 = np.array([dzl[l, 0] * self.W3P[0, l >> 4] for l in range(64)])
            dz2 = dh2 * (1 - h2 ** 2)
        self.scatter(TB, dz2)
        dacc = self.chain(self.W2B, dz2, [np.zeros((64, 4)) for _ in range(4)])
        if self.HM:
            aop = np.array([DOT[l & 15, 4 * (l >> 4):4 * (l >> 4) + 4] for l in range(64)])
            for T in range(4):
                hF = self.gather(TA, T)
                for e in range(4):
                    self.gW3[T] = mfma16(aop[:, e], hF[:, e], self.gW3[T])
        aT = [self.gather32(TB, Ti) for Ti in range(2)]
        for Ti in range(2):
            self.gb2w[Ti] += aT[Ti].sum(1)
        dz1 = np.concatenate(dacc, axis=1) * (1 - h1 ** 2)
        self.scatter(TA, h1)
        self.scatter(TB, dz1)
        for Tj in range(2):
            bT = self.gather32(TA, Tj)
            for Ti in range(2):
                self.gW2w[Ti, Tj] = mfma32x16(aT[Ti], bT, self.gW2w[Ti, Tj])
        cF = [self.gather(TB, T) for T in range(4)]
        for T in range(4):
            self.gb1[T] += cF[T].sum(1)
        for lane in range(64):
            for e in range(XE):
                TA[XE * (lane >> 4) + e, lane & 15] = x[lane, e]
        for Tj in range(self.XT):
            xF = self.gather(TA, Tj)
            xF = np.where(((16 * Tj + (LANES & 15)) < 4 * self.KS1)[:, None], xF, 0.0)
            for e in range(4):
                for Ti in range(4):
                    self.gW1[Ti, Tj] = mfma16(cF[Ti][:, e], xF[:, e], self.gW1[Ti, Tj])
        for c in range(self.XR):
            xr = np.array([TA[16 * self.XT + c, 4 * (l >> 4):4 * (l >> 4) + 4] for l in range(64)])
            for Ti in range(4):
                self.gW1r[Ti, c] += (cF[Ti] * xr).sum(1)

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
                    for Tj in range(self.XT):
                        if 16 * Tj + s < O:
                            G[oW1 + row * O + 16 * Tj + s] += self.gW1[Ti, Tj, lane, r]
                    if self.HM and 4 * g + r < self.nout:
                        G[oW3 + (4 * g + r) * 64 + 16 * Ti + s] += self.gW3[Ti, lane, r]
            for Ti in range(2):
                for Tj in range(2):
                    for r in range(16):
                        row = 32 * Ti + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3)
                        G[oW2 + row * 64 + 32 * Tj + (lane & 31)] += self.gW2w[Ti, Tj, lane, r]
        for Ti in range(4):
            v1 = sum_groups(self.gb1[Ti])
            for lane in range(16):
                G[ob1 + 16 * Ti + lane] += v1[lane]
            for c in range(self.XR):
                vr = sum_groups(self.gW1r[Ti, c])
                if 16 * self.XT + c < O:
                    for lane in range(16):
                        G[oW1 + (16 * Ti + lane) * O + 16 * self.XT + c] += vr[lane]
        for Ti in range(2):
            v2 = self.gb2w[Ti] + self.gb2w[Ti][LANES ^ 32]
            for lane in range(32):
                G[ob2 + 32 * Ti + lane] += v2[lane]
        if self.HM:
            for r in range(self.NS):
                vb, vs = row_sum(self.hb[:, r]), row_sum(self.hsg[:, r])
                for g in range(4):
                    if g + 4 * r < self.nout:
                        G[ob3 + g + 4 * r] += vb[16 * g]
                        G[oLs + g + 4 * r] += vs[16 * g]
        else:
            for q in range(16):
                v = row_sum(self.gW3s[:, q])
                for g in range(4):
                    G[oW3 + feat16(q, g)] += v[16 * g]
            G[ob3] += row_sum(self.hb[:, 0])[0]
            if self.actor:
                G[oLs] += row_sum(self.hsg[:, 0])[0]
        G[P + 0], G[P + 1], G[P + 2], G[P + 5] = (self.stats[k].sum() for k in range(4))
        return G, P


def emulate_grad16f(params, O, A, actor, data, n, norm=None, waves=3):
    ws = [WaveF16(params, O, A, actor, norm) for _ in range(waves)]
    for tile in range((n + 15) // 16):
        ws[tile % waves].tile(tile * 16, n, data)
    total, P = None, None
    for w in ws:
        G, P = w.fold()
        total = G if total is None else total + G
    return total, P
