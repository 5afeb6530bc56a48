"""CPU model of the quantised integer body on BASELINE configs[1]'s data (numpy only): how many records / rows a query lists under
the triangle-inequality slack, and how often a stream of L tiles cannot be certified.  (The numbers DESIGN.md 4.1 quotes.)"""
import numpy as np
rng = np.random.default_rng(0)
nq, nt = 2000, 10016
Q, T = rng.random((nq, 128), dtype=np.float32), rng.random((nt, 128), dtype=np.float32)
lo, s = 0.0, 1.0 / 255
kq, kt = np.rint((Q - lo) / s), np.rint((T - lo) / s)
E = (np.linalg.norm(Q - (lo + s * kq), axis=1) + np.linalg.norm(T - (lo + s * kt), axis=1).max())[:, None]
D = (kq ** 2).sum(1)[:, None] + (kt ** 2).sum(1)[None, :] - 2 * kq @ kt.T
dhat = np.sqrt(D) * s
Qd, Td = Q.astype(np.float64), T.astype(np.float64)
dtrue = np.sqrt(((Qd ** 2).sum(1)[:, None] + (Td ** 2).sum(1)[None, :] - 2 * Qd @ Td.T).clip(0))
print("s %.5f  mean E/s %.2f  max |d - dhat| / s %.2f" % (s, E.mean() / s, np.abs(dtrue - dhat).max() / s))
d2 = np.sort(dtrue, axis=1)[:, 1][:, None]
ntile = nt // 32
rows = np.arange(32); e = rows // 16; h = (rows // 4) % 2           # a record = (tile, e, h): the 8 rows 16 e + 8 (r >> 2) + 4 h + (r & 3)
dh = dhat.reshape(nq, ntile, 32)
rec = np.stack([dh[:, :, (e == ee) & (h == hh)].min(2) for ee in (0, 1) for hh in (0, 1)], axis=2)
a2 = np.sort(rec.reshape(nq, -1), axis=1)[:, 1][:, None]
print("records listed per query (blind bound, two slacks) %.2f; against the known second distance (one slack) %.2f" %
      ((rec.reshape(nq, -1) <= a2 + 2 * E).sum(1).mean(), (rec.reshape(nq, -1) <= d2 + E).sum(1).mean()))
for L in (128, 64, 32, 16):
    fails = 0
    for t0 in range(0, ntile, L):
        for hh in (0, 1):
            blk = rec[:, t0:t0 + L, [hh, 2 + hh]].reshape(nq, -1)
            if blk.shape[1] >= 3:
                fails += ((np.partition(blk, 2, axis=1)[:, 2:3] - E) <= d2 * (1 + 1e-5)).sum()
    print("substreams of %3d tiles: %.4f uncertified streams per query" % (L, fails / nq))
