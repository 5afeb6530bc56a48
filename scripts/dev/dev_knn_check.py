"""Dev-only: drive the KNN entry points directly through ctypes (no binding table) and diff vs the oracle."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
L = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "sfm_mvs_amd/lib/libsfmhip.so"))
L.sfm_knn2_l2_f32_ws_bytes.restype = C.c_size_t
L.sfm_last_error.restype = C.c_char_p
vp = C.c_void_p

def run(q, t, iters=0):
    nq, nt = q.shape[0], t.shape[0]
    idx = torch.empty((nq, 2), dtype=torch.int32, device="cuda"); dist = torch.empty((nq, 2), dtype=torch.float32, device="cuda")
    stats = torch.zeros(4, dtype=torch.int32, device="cuda")
    need = L.sfm_knn2_l2_f32_ws_bytes(C.c_int64(nq), C.c_int64(nt), 128)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    def call():
        rc = L.sfm_knn2_l2_f32(vp(q.data_ptr()), C.c_int64(nq), C.c_int64(q.stride(0)), vp(t.data_ptr()), C.c_int64(nt), C.c_int64(t.stride(0)),
                               128, vp(idx.data_ptr()), vp(dist.data_ptr()), vp(stats.data_ptr()), vp(ws.data_ptr()), C.c_size_t(need), None)
        assert rc == 0, L.sfm_last_error()
    call(); torch.cuda.synchronize()
    ms = None
    if iters:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): call()
        e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / iters
    return idx.cpu().numpy(), dist.cpu().numpy(), stats.cpu().numpy(), ms

def sift_like(rng, n):
    d = np.abs(rng.standard_normal((n, 128))) ** 2
    d /= np.linalg.norm(d, axis=1, keepdims=True); d = np.minimum(d, 0.2); d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.clip(np.rint(d * 512), 0, 255).astype(np.float32)

rng = np.random.default_rng(0)
for name, nq, nt, gen in [("uniform", 777, 1234, None), ("uniform", 3000, 2500, None), ("sift", 2000, 3001, "sift"), ("tiny", 5, 3, None), ("one", 40, 1, None), ("dups", 300, 600, "dups")]:
    if gen == "sift":
        qn = sift_like(rng, nq); tn = sift_like(rng, nt)
        k = nq // 3; perm = rng.permutation(nt)[:k]; tn[perm] = np.clip(qn[:k] + np.rint(rng.normal(0, 2, (k, 128))), 0, 255)
    elif gen == "dups":
        qn = rng.random((nq, 128), dtype=np.float32); base = rng.random((nt // 6, 128), dtype=np.float32); tn = np.tile(base, (6, 1))
    else:
        qn = rng.random((nq, 128), dtype=np.float32); tn = rng.random((nt, 128), dtype=np.float32)
    q = torch.from_numpy(qn).cuda(); t = torch.from_numpy(tn).cuda()
    idx, dist, stats, _ = run(q, t)
    oi, od = O.knn2(qn, tn, nthreads=8)
    print(f"{name} {nq}x{nt}: idx_equal={np.array_equal(idx, oi)} dist_bitequal={np.array_equal(dist.view(np.uint32), od.view(np.uint32))} "
          f"mismatch_rows={(idx != oi).any(1).sum()} stats={stats}", flush=True)

for nq, nt in [(10000, 10000), (4096, 4096), (50000, 50000)]:
    q = torch.rand((nq, 128), device="cuda"); t = torch.rand((nt, 128), device="cuda")
    idx, dist, stats, ms = run(q, t, iters=10)
    print(f"perf {nq}x{nt}: {ms:.4f} ms/call  {nq*nt/ms*1e3:.3e} dist/s  {nq*nt*256/ms/1e9:.1f} TFLOP/s(GEMM-form)  stats={stats}", flush=True)
    if nq <= 10000:
        oi, od = O.knn2(q.cpu().numpy(), t.cpu().numpy(), nthreads=8)
        print("   parity idx", np.array_equal(idx, oi), "dist", np.array_equal(dist.view(np.uint32), od.view(np.uint32)))
