"""How the sixteen last-round multipliers of the attention dropout's 4 x 4 block generator (STONK_C2_BLK, csrc/common.h)
were chosen: the first hash round of a block is a 24-bit word y, member i is dropped iff lo32(y * C[i]) < p * 2^32. Over
ALL 2^24 values of y (the exact joint distribution when y is uniform) a greedy search over random odd 24-bit candidates
keeps the set whose worst pairwise (and, half-weighted, triple) joint drop rate is closest to p^2 (p^3) at p = 0.1 and
0.25; the last lines print the chosen set's worst pair / triple / quadruple ratios and the distribution of the number of
drops per block against the binomial. CPU only, about ten minutes."""
import numpy as np, itertools
from math import comb
y = np.arange(1 << 24, dtype=np.uint64)
rng = np.random.default_rng(11)
N = float(1 << 24)
def drops(c, p):
    thr = np.uint64(int(p * 2**32 + 0.5))
    return np.packbits(((y * np.uint64(c)) & np.uint64(0xFFFFFFFF)) < thr).view(np.uint64)
def pc(x): return int(np.bitwise_count(x).sum())
cands = sorted({int(rng.integers(1 << 22, 1 << 24)) | 1 for _ in range(120)})
ps = (0.1, 0.25)
D = {p: {c: drops(c, p) for c in cands} for p in ps}
chosen = [cands[5]]
while len(chosen) < 16:
    best, bc = 1e9, None
    for c in cands:
        if c in chosen: continue
        w = 0
        for p in ps:
            dc = D[p][c]
            for a in chosen:
                w = max(w, abs(pc(D[p][a] & dc) / N / p**2 - 1))
            if w > best: break
            for a, b in itertools.combinations(chosen, 2):
                w = max(w, 0.5 * abs(pc(D[p][a] & D[p][b] & dc) / N / p**3 - 1))
                if w > best: break
            if w > best: break
        if w < best: best, bc = w, c
    chosen.append(bc)
    print(len(chosen), hex(bc), f"score {best:.4f}", flush=True)
print([hex(c) for c in chosen])
for p in ps:
    wp = max(abs(pc(D[p][a] & D[p][b]) / N / p**2 - 1) for a, b in itertools.combinations(chosen, 2))
    wt = max(abs(pc(D[p][a] & D[p][b] & D[p][c]) / N / p**3 - 1) for a, b, c in itertools.combinations(chosen, 3))
    wq = max(abs(pc(D[p][a] & D[p][b] & D[p][c] & D[p][d]) / N / p**4 - 1) for a, b, c, d in itertools.combinations(chosen[:10], 4))
    cnt = np.zeros(1 << 24, dtype=np.int8)
    for c in chosen: cnt += np.unpackbits(D[p][c].view(np.uint8)).astype(np.int8)
    obs = np.bincount(cnt, minlength=17)[:17] / N
    exp = np.array([comb(16, k) * p**k * (1-p)**(16-k) for k in range(17)])
    print(p, "worst pair", round(wp, 4), "worst triple", round(wt, 4), "worst quad(first 10)", round(wq, 3), "count dist ratio", np.round(obs[:10] / exp[:10], 3))
