"""BASELINE.json's configurations at their FULL sizes on the MI355X: size-independent properties at full size plus an
oracle comparison on a slice of the same problem (config 4: 500 cameras x 200 000 points), a 50 000 x 50 000 SIFT-like
pair bit-exact against the oracle and an image sequence of config 5's shape, and SIFT on the reference's own sample
photograph (tests/golden/photo_gray.npz) at 1936 x 1296 and at the reference's working size 968 x 648."""
import os

import numpy as np
import pytest
import torch

from datagen import GOLDEN, load_pose_csv, ring_cameras

pytestmark = pytest.mark.gpu


def test_config4_dense_sweep_500_cameras_x_200k_points(hip, oracle):
    """1e8 observations (800 MB of float32 pixels streamed once per sweep): deterministic, additive over camera subsets,
    and — on a 500 x 2000 slice of the SAME problem — equal to the oracle; the full sweep's per-point blocks of those
    2000 points must equal the slice's (a point's blocks depend on its own observations only)."""
    ncam, npt, dev = 500, 200_000, torch.device("cuda")
    K, _ = load_pose_csv()
    g = torch.Generator(device="cpu").manual_seed(3)
    cams = torch.from_numpy(ring_cameras(ncam)).to(dev)
    X = torch.randn((npt, 3), generator=g)
    X = (X / X.norm(dim=1, keepdim=True).clamp(min=1.0) * torch.rand((npt, 1), generator=g).clamp(min=0.2)).to(dev)
    obs = torch.empty((ncam, npt, 2), dtype=torch.float32, device=dev)
    zero = torch.zeros((npt, 2), device=dev)
    for c in range(ncam):
        obs[c] = hip.project_residual(cams[c:c + 1], K, X, zero, want_proj=True)["proj"]
    obs += 0.5 * torch.randn(obs.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(4))
    cams_p = cams * (1 + 0.01 * torch.randn(cams.shape, device=dev, dtype=torch.float64, generator=torch.Generator(device=dev).manual_seed(5)))
    full = hip.ba_dense_sweep(cams_p, K, X, obs)
    again = hip.ba_dense_sweep(cams_p, K, X, obs)
    for k in full:
        assert torch.equal(full[k], again[k]), k                                      # fixed-order reductions
    assert torch.isfinite(full["JtJ_cam"]).all() and torch.isfinite(full["JtJ_pt"]).all()
    a = hip.ba_dense_sweep(cams_p[:250], K, X, obs[:250].contiguous())
    b = hip.ba_dense_sweep(cams_p[250:], K, X, obs[250:].contiguous())
    assert torch.allclose(full["JtJ_cam"][:250], a["JtJ_cam"], rtol=1e-12, atol=0) and torch.allclose(full["JtJ_cam"][250:], b["JtJ_cam"], rtol=1e-12, atol=0)
    assert torch.allclose(full["JtJ_pt"], a["JtJ_pt"] + b["JtJ_pt"], rtol=1e-10, atol=1e-6)
    assert torch.allclose(full["Jtr_pt"], a["Jtr_pt"] + b["Jtr_pt"], rtol=1e-9, atol=1e-6)
    assert full["sumsq"].item() == pytest.approx(a["sumsq"].item() + b["sumsq"].item(), rel=1e-12)
    # metric sanity: sigma = 0.5 px noise on 1 % perturbed cameras -> a finite, large but bounded cost
    assert 2 * ncam * npt * 0.2 < full["sumsq"].item() < 2 * ncam * npt * 1e4
    # oracle on a slice of the same problem
    ns = 2000
    Xs, obs_s = X[:ns].contiguous(), obs[:, :ns].contiguous()
    ci = np.repeat(np.arange(ncam), ns).astype(np.int32)
    pi = np.tile(np.arange(ns), ncam).astype(np.int32)
    want = oracle.project_residual(cams_p.cpu().numpy(), K, Xs.cpu().numpy(), obs_s.cpu().numpy().reshape(-1, 2), ci, pi)
    got = hip.ba_dense_sweep(cams_p, K, Xs, obs_s)
    for key in ("JtJ_cam", "Jtr_cam", "JtJ_pt", "Jtr_pt"):
        gv, wv = got[key].cpu().numpy(), want[key]
        assert np.abs(gv - wv).max() <= 1e-10 * np.abs(wv).max(), key
    assert got["sumsq"].item() == pytest.approx(want["sumsq"][0], rel=1e-12)
    for key in ("JtJ_pt", "Jtr_pt"):       # (the camera loop is tiled differently at the two sizes: same sums, another order)
        assert (full[key][:ns] - got[key]).abs().max().item() <= 1e-12 * got[key].abs().max().item(), key


def _sift_like_dev(n, g, dev):
    d = torch.randn((n, 128), generator=g, device=dev).abs_().square_()
    d /= d.norm(dim=1, keepdim=True)
    d = torch.minimum(d, torch.tensor(0.2, device=dev))
    d /= d.norm(dim=1, keepdim=True)
    return (d * 512).round_().clamp_(0, 255)


def test_config5_pair_50k_x_50k_bit_exact_and_sequence_of_four(hip, oracle):
    """Config 5's unit of work at full size: one 50 000 x 50 000 pair of SIFT-like descriptors (2.5e9 distances)
    bit-identical to the oracle's direct-form scan on every host core, Lowe mask included; then a 4-image sequence
    through PairPipeline (pairs in flight on separate streams) in which every planted match must come back as the
    nearest neighbour and survive the ratio test."""
    dev, n = torch.device("cuda"), 50_000
    g = torch.Generator(device=dev).manual_seed(100)
    imgs = [_sift_like_dev(n, g, dev)]
    planted = []
    for k in range(1, 4):
        nxt = _sift_like_dev(n, g, dev)
        src = torch.randperm(n, generator=g, device=dev)[: int(0.3 * n)]
        dst = torch.randperm(n, generator=g, device=dev)[: int(0.3 * n)]
        nxt[dst] = (imgs[-1][src] + torch.randn((len(src), 128), generator=g, device=dev).mul_(2).round_()).clamp_(0, 255)
        imgs.append(nxt)
        planted.append((src, dst.to(torch.int32)))
    idx, dist = hip.knn2(imgs[0], imgs[1])
    oq, ot, cnt = hip.ratio_compact(idx, dist, 0.70)
    wi, wd = oracle.knn2(imgs[0].cpu().numpy(), imgs[1].cpu().numpy(), nthreads=os.cpu_count() or 8)
    assert np.array_equal(idx.cpu().numpy(), wi) and np.array_equal(dist.cpu().numpy(), wd)
    wq, wt, _ = oracle.ratio_filter(wi, wd, 0.70)
    m = int(cnt.item())
    assert m == len(wq) and np.array_equal(oq[:m].cpu().numpy(), wq) and np.array_equal(ot[:m].cpu().numpy(), wt)
    assert m >= int(0.29 * n)
    pipe = hip.PairPipeline(n, n, dev, ratio=0.70, depth=3)
    keep = []
    for k in range(3):
        slot, st, (pidx, pdist, poq, pot, pcnt) = pipe.submit(imgs[k], imgs[k + 1], after=False)
        with torch.cuda.stream(st):
            keep.append((pidx.clone(), poq.clone(), pot.clone(), pcnt.clone()))
    pipe.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(keep[0][0], idx)                                               # the pipelined pair 0 = the plain call
    for k in range(3):
        pidx, poq, pot, pcnt = keep[k]
        src, dst = planted[k]
        assert torch.equal(pidx[src.long(), 0], dst)                                   # every planted match is the nearest neighbour
        mk = int(pcnt.item())
        survived = torch.zeros(n, dtype=torch.bool, device=dev)
        survived[poq[:mk].long()] = True
        assert survived[src.long()].float().mean().item() > 0.999                     # ... and passes the 0.70 ratio test
        assert torch.all(poq[1:mk] > poq[:mk - 1])                                    # ascending queryIdx


@pytest.mark.parametrize("downscaled", [False, True])
def test_sift_on_the_references_sample_photograph(hip, oracle, downscaled):
    """image.jpg of the reference (a real 1936 x 1296 photograph; grey fixture made by tests/golden/make_golden.py):
    keypoints and descriptors bit-identical to the oracle at full size and after the reference's one pyrDown
    (sfm.py:40 -> 968 x 648, its working size)."""
    from sfm_mvs_amd import sift
    gray = np.load(os.path.join(GOLDEN, "photo_gray.npz"))["gray"]
    assert gray.shape == (1296, 1936) and gray.dtype == np.uint8
    if downscaled:
        small = oracle.pyrdown(gray)
        got_small = sift.pyrdown(torch.as_tensor(gray).cuda()).cpu().numpy()
        assert small.shape == (648, 968) and np.array_equal(small, got_small)
        gray = small
    h, w = gray.shape
    kpo, deso = oracle.sift(gray)
    eng = sift.Sift(w, h, "cuda", max_keypoints=1 << 17)
    kp, des = eng.run(torch.as_tensor(gray).cuda())
    kp, des = kp.cpu().numpy(), des.cpu().numpy()
    assert len(kpo) > 1000 and kp.shape == kpo.shape
    assert np.array_equal(kp.view(np.int32), kpo.view(np.int32)) and np.array_equal(des, deso)
    assert np.all(des == np.rint(des)) and des.max() <= 255 and abs(np.linalg.norm(des, axis=1).mean() - 512) < 20


def test_config1_surrogate_two_views_of_the_photograph(hip, oracle):
    """BASELINE config 1 is "the first two Gustav images"; the dataset is not available, the reference ships ONE photograph.
    Two overlapping 968 x 648 windows of its pyrDown-ed half-size frame... are not enough pixels, so the windows are cut
    from the full frame and halved as sfm.py:40 does: the second view is the first displaced by a known (dx, dy).  Through
    the reference's own entry point find_features(img0, img1) (sfm.py:242: BGR frames in, pts0 / pts1 out), HIP end to end,
    against the same call over the oracle's cv2 facade (bit-identical: SIFT, KNN, ratio and gather all are), and against
    the known displacement."""
    from oracle_backend import oracle_pipeline_backend
    from sfm_mvs_amd import pipeline as pl
    gray = np.load(os.path.join(GOLDEN, "photo_gray.npz"))["gray"]
    dx, dy = 148, 64                                            # full-resolution displacement (even: exact after pyrDown)
    h, w = 1296 - dy, 1936 - dx
    h, w = h - h % 2, w - w % 2
    v0 = gray[:h, :w]
    v1 = gray[dy:dy + h, dx:dx + w]
    frames = [np.ascontiguousarray(np.repeat(pl.img_downscale(v, 2)[:, :, None], 3, axis=2)) for v in (v0, v1)]   # BGR, half size
    p0, p1 = pl.find_features(frames[0], frames[1])
    q0, q1 = pl.find_features(frames[0], frames[1], be=oracle_pipeline_backend(oracle))
    assert p0.shape == p1.shape == q0.shape and p0.shape[0] > 300
    assert np.array_equal(p0, q0) and np.array_equal(p1, q1)
    d = p0 - p1                                                 # a point of view 0 sits (dx, dy) / 2 further right / down than in view 1
    good = (np.abs(d[:, 0] - dx / 2) < 1.0) & (np.abs(d[:, 1] - dy / 2) < 1.0)
    assert good.mean() > 0.9, good.mean()
