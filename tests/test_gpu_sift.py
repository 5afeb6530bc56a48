"""SIFT + preprocessing on the MI355X: libsfmhip.so through its C-ABI against the CPU oracle on the same
images — keypoints and descriptors bit for bit (the kernels keep the sequential algorithm's float32 operation order) —
plus an images -> keypoints -> matches round trip through the KNN path."""
import numpy as np
import pytest
import torch

from datagen import scene_image

pytestmark = pytest.mark.gpu


def _sift_hip(g, **kw):
    from sfm_mvs_amd import sift
    h, w = g.shape
    eng = sift.Sift(w, h, "cuda", **kw)
    kp, des = eng.run(torch.as_tensor(g).cuda())
    return kp.cpu().numpy(), des.cpu().numpy(), eng


def _same(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.int32), b.view(np.int32))


def test_bgr2gray_and_pyrdown_bit_exact(hip, oracle):
    from sfm_mvs_amd import sift
    rng = np.random.default_rng(0)
    for (h, w) in [(1, 1), (7, 5), (64, 64), (217, 333), (648, 968)]:
        bgr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        d = torch.as_tensor(bgr).cuda()
        assert np.array_equal(sift.bgr2gray(d).cpu().numpy(), oracle.bgr2gray(bgr))
        assert np.array_equal(sift.pyrdown(d).cpu().numpy(), oracle.pyrdown(bgr))
        gray = np.ascontiguousarray(bgr[..., 1])
        assert np.array_equal(sift.pyrdown(torch.as_tensor(gray).cuda()).cpu().numpy(), oracle.pyrdown(gray))


@pytest.mark.parametrize("w,h,seed", [(160, 120, 0), (256, 192, 1), (333, 217, 2), (97, 301, 5), (640, 480, 7)])
def test_sift_bit_exact_vs_oracle(hip, oracle, w, h, seed):
    g = scene_image(w, h, seed)
    kpo, deso = oracle.sift(g)
    kp, des, eng = _sift_hip(g)
    assert len(kpo) > 100
    assert _same(kp, kpo) and _same(des, deso)
    n, raw, cand, _ = eng.count.tolist()
    assert n == len(kpo) and raw >= n and cand <= raw


def test_sift_small_flat_and_strided(hip, oracle):
    from sfm_mvs_amd import sift
    flat = np.full((40, 56), 93, np.uint8)
    kp, des, _ = _sift_hip(flat)
    assert kp.shape == (0, 8) and des.shape == (0, 128)
    tiny = scene_image(24, 17, 3)
    kpo, deso = oracle.sift(tiny)
    kp, des, _ = _sift_hip(tiny)
    assert _same(kp, kpo) and _same(des, deso)
    # a view with a row stride larger than its width (a crop of a bigger frame), no copy
    big = torch.as_tensor(scene_image(300, 200, 9)).cuda()
    crop = big[20:180, 50:250]
    eng = sift.Sift(200, 160, "cuda")
    kp, des = eng.run(crop)
    kpo, deso = oracle.sift(crop.cpu().numpy())
    assert _same(kp.cpu().numpy(), kpo) and _same(des.cpu().numpy(), deso)


def test_sift_other_parameters(hip, oracle):
    g = scene_image(240, 180, 12)
    for kw, okw in [(dict(n_octave_layers=4, contrast_threshold=0.03), dict(n_octave_layers=4, contrast=0.03)),
                    (dict(edge_threshold=6.0, sigma=1.4), dict(edge=6.0, sigma=1.4)),
                    (dict(n_octave_layers=2), dict(n_octave_layers=2))]:
        kpo, deso = oracle.sift(g, **okw)
        kp, des, _ = _sift_hip(g, **kw)
        assert len(kpo) > 50 and _same(kp, kpo) and _same(des, deso)


def test_sift_errors_are_loud(hip):
    from sfm_mvs_amd import SfmHipError, sift
    g = torch.as_tensor(scene_image(160, 120, 0)).cuda()
    with pytest.raises(SfmHipError):
        sift.Sift(160, 120, "cuda", max_keypoints=64).run(g)          # capacity exceeded: reported, not truncated silently
    with pytest.raises(SfmHipError):
        sift.Sift(160, 120, "cuda").run(g.cpu())                       # host memory
    with pytest.raises(SfmHipError):
        sift.Sift(160, 120, "cuda").run(g[:, :100])                    # wrong size
    with pytest.raises(SfmHipError):
        sift.Sift(160, 120, "cuda", sigma=8.0).run(g)                  # needs more than 55 taps


def test_sift_repeatable_and_reusable(hip):
    from sfm_mvs_amd import sift
    eng = sift.Sift(256, 192, "cuda")
    a = torch.as_tensor(scene_image(256, 192, 1)).cuda()
    b = torch.as_tensor(scene_image(256, 192, 2)).cuda()
    kpa, dea = (t.clone() for t in eng.run(a))
    kpb, _ = eng.run(b)
    assert len(kpb) != len(kpa) or not torch.equal(kpa, kpb)
    kpa2, dea2 = eng.run(a)
    assert torch.equal(kpa.view(torch.int32), kpa2.view(torch.int32)) and torch.equal(dea, dea2)


def test_images_to_matches_round_trip(hip):
    """find_features (sfm.py:242-270) from pixels: SIFT on two overlapping views of one scene, BF-KNN k=2 + Lowe 0.70 on
    the descriptors, gather of the keypoint coordinates; the matches must reproduce the known translation."""
    from sfm_mvs_amd import cv2compat as cv2
    dx, dy = 23.0, -11.0
    g0 = scene_image(320, 240, 21)
    g1 = scene_image(320, 240, 21, shift=(dx, dy))          # content moves by (-dx, -dy)
    s = cv2.xfeatures2d.SIFT_create()
    kp0, des0 = s.detectAndCompute(g0, None)
    kp1, des1 = s.detectAndCompute(g1, None)
    assert len(kp0) > 500 and des0.shape == (len(kp0), 128) and isinstance(kp0[0].pt, tuple)
    matches = cv2.BFMatcher().knnMatch(des0, des1, k=2)
    good = [m for m, n in matches if m.distance < 0.70 * n.distance]
    assert len(good) > 150
    p0 = np.float32([kp0[m.queryIdx].pt for m in good])
    p1 = np.float32([kp1[m.trainIdx].pt for m in good])
    d = p0 - p1
    ok = (np.abs(d[:, 0] - dx) < 0.5) & (np.abs(d[:, 1] - dy) < 0.5)
    assert ok.mean() > 0.95
    # the reference's own entry point: find_features(img0, img1) -> pts0, pts1 (sfm.py:242), from BGR frames, on the device
    from sfm_mvs_amd import pipeline as pl
    q0, q1 = pl.find_features(np.repeat(g0[:, :, None], 3, 2), np.repeat(g1[:, :, None], 3, 2))
    assert q0.dtype == np.float32 and q0.shape == q1.shape == (len(good), 2)
    assert np.array_equal(q0, p0) and np.array_equal(q1, p1)                          # same matches, same (queryIdx) order
    r0, r1 = pl.find_features(g0, g1)                                                  # already-grey frames are accepted too
    assert np.array_equal(r0, p0) and np.array_equal(r1, p1)


def test_cv2compat_preprocessing(hip, oracle):
    from sfm_mvs_amd import cv2compat as cv2
    bgr = np.random.default_rng(3).integers(0, 256, (90, 130, 3), dtype=np.uint8)
    small = cv2.pyrDown(bgr)
    assert small.shape == (45, 65, 3) and np.array_equal(small, oracle.pyrdown(bgr))
    gray = cv2.cvtColor(small, cv2.COLOR_BGR2GRAY)
    assert gray.shape == (45, 65) and np.array_equal(gray, oracle.bgr2gray(small))


def test_driver_from_pixels_recovers_known_geometry(hip):
    """sfm.py end to end from images: img_downscale(…, 2) -> cvtColor -> SIFT -> matcher -> essential matrix / PnP /
    triangulation, on a rendered sequence with known cameras (translation along x, three depth layers)."""
    from datagen import decompose_P, layered_views
    from sfm_mvs_amd import pipeline as pl
    images, K, P = layered_views(5, 400, 300, 4)
    big = [np.repeat(np.repeat(im, 2, axis=0), 2, axis=1) for im in images]      # frames twice the working size
    small = [pl.img_downscale(b, 2) for b in big]
    assert small[0].shape == images[0].shape
    feats = pl.features_from_images(images)
    assert all(len(k) > 1500 and d.shape == (len(k), 128) for k, d in feats)
    out = pl.run_sfm(feats, K, images=images)
    pose = out["posearr"][9:].reshape(-1, 3, 4)
    assert len(pose) == 5
    for k, Pk in enumerate(pose):
        R, t = decompose_P(K, Pk)
        assert np.abs(R - np.eye(3)).max() < 5e-3
        assert np.abs(-R.T @ t - np.array([k, 0.0, 0.0])).max() < 3e-2          # unit first baseline (recoverPose)
    assert max(out["errors"]) < 1.0
    feats_dev = pl.features_from_images(images, on_device=True)                # features kept in HBM: same values, same run
    assert all(torch.is_tensor(k) and k.is_cuda and np.array_equal(k.cpu().numpy(), kh) and np.array_equal(d.cpu().numpy(), dh)
               for (k, d), (kh, dh) in zip(feats_dev, feats))
    assert np.array_equal(pl.run_sfm(feats_dev, K, images=images)["posearr"], out["posearr"])
    out_px = pl.run_sfm_images(big, K, downscale=2)                            # pyrDown'ed frames: other pixels, same scene
    for k, Pk in enumerate(out_px["posearr"][9:].reshape(-1, 3, 4)):
        R, t = decompose_P(K, Pk)
        assert np.abs(R - np.eye(3)).max() < 1e-2 and np.abs(-R.T @ t - np.array([k, 0.0, 0.0])).max() < 6e-2
    Z = out["Xtot"][1:, 2] * 0.25                                              # metric scale: the true baseline is 0.25
    near = [np.mean(np.abs(Z - d) < 0.15) for d in (10.0, 6.0, 4.0)]
    assert sum(near) > 0.9 and min(near) > 0.05                                # the cloud sits on the three layers


def test_sift_full_size_photograph_shape(hip, oracle):
    """The reference's photographs are 1936 x 1296 before img_downscale; SIFT at that size (10 octaves, 3872 x 2592 base)
    must still agree with the oracle bit for bit.  Content: a tiled procedural scene (generation cost, not coverage)."""
    tile = scene_image(484, 324, 31)
    g = np.ascontiguousarray(np.tile(tile, (4, 4)))
    assert g.shape == (1296, 1936)
    kpo, deso = oracle.sift(g)
    kp, des, eng = _sift_hip(g, max_keypoints=1 << 18)
    assert len(kpo) > 20000 and _same(kp, kpo) and _same(des, deso)


def test_sift_pipeline_matches_single_engine(hip):
    """Frames in flight on several streams (SiftPipeline) give exactly what one engine gives frame by frame."""
    from sfm_mvs_amd import sift
    frames = [torch.as_tensor(scene_image(288, 208, 40 + k)).cuda() for k in range(7)]
    one = sift.Sift(288, 208, "cuda")
    want = []
    for f in frames:
        kp, des = one.run(f)
        want.append((kp.clone(), des.clone()))
    pipe = sift.SiftPipeline(288, 208, "cuda", depth=3)
    got, pending = [], []
    for f in frames:
        if len(pending) == pipe.depth:
            st, eng = pending.pop(0)
            st.synchronize()
            n = eng.check_capacity()
            got.append((eng.keypoints[:n].clone(), eng.descriptors[:n].clone()))
        _, st, eng = pipe.submit(f)
        pending.append((st, eng))
    for st, eng in pending:
        st.synchronize()
        n = eng.check_capacity()
        got.append((eng.keypoints[:n].clone(), eng.descriptors[:n].clone()))
    assert len(got) == len(want)
    for (ka, da), (kb, db) in zip(got, want):
        assert torch.equal(ka.view(torch.int32), kb.view(torch.int32)) and torch.equal(da, db)


def test_feature_stream_is_the_serial_order_and_fails_loudly(hip):
    """pipeline.FeatureStream (features produced ahead of the driver by a second host thread): every frame's keypoints and
    descriptors equal features_from_images' bit for bit, random access blocks until the frame is there, and a frame the kernels
    reject surfaces in the CONSUMER as the library's error instead of hanging it."""
    from datagen import scene_image
    from sfm_mvs_amd import SfmHipError
    from sfm_mvs_amd import pipeline as pl
    frames = [np.stack([scene_image(320, 240, 40 + k)] * 3, -1) for k in range(7)]
    big = [np.repeat(np.repeat(f, 2, axis=0), 2, axis=1) for f in frames]
    want = pl.features_from_images([pl.img_downscale(b, 2) for b in big], on_device=True)
    fs = pl.FeatureStream(big, 2, depth=2, lookahead=3)
    kp5, des5 = fs[5]                                              # out of order: blocks until frame 5 exists
    assert torch.equal(kp5, want[5][0]) and torch.equal(des5, want[5][1])
    for k, (kp, des) in enumerate(fs):
        assert torch.equal(kp, want[k][0]) and torch.equal(des, want[k][1]), k
    fs.close()
    assert all(s is not None and tuple(s.shape) == (240, 320, 3) for s in fs.small)
    bad = list(big)
    bad[3] = torch.from_numpy(big[3].astype(np.float32))           # a tensor frame that is not uint8: pyrdown refuses
    fs = pl.FeatureStream(bad, 2)
    with pytest.raises(SfmHipError):
        fs[6]
    fs.close()


def test_feature_stream_abandoned_mid_sequence_does_not_hang(hip):
    """A consumer that gives up (an exception in the driver) closes the stream: the producer stops instead of waiting for the
    consumer to catch up, the cached pipelines go back, and the next run gets them."""
    from datagen import scene_image
    from sfm_mvs_amd import pipeline as pl
    big = [np.stack([np.repeat(np.repeat(scene_image(160, 120, 60 + k), 2, 0), 2, 1)] * 3, -1) for k in range(20)]
    fs = pl.FeatureStream(big, 2, depth=2, lookahead=3)
    _ = fs[1]
    fs.close()                                                     # frames 6.. were never asked for
    assert not fs._thread.is_alive()
    with pytest.raises(Exception):
        fs[19]
    fs2 = pl.FeatureStream(big, 2, depth=2, lookahead=3)
    assert len(list(fs2)) == 20
    fs2.close()
    assert not pl._SIFT_PIPES_BUSY


def test_independent_streams_probe(hip):
    """ops.independent_streams: the streams it returns run beside each other (ops.streams_overlap, both directions) — what the
    from-pixels job needs of its chain stream and feature streams (a chain stream that shares a hardware queue with a feature
    stream costs the job 20 %: profiles/r05_stream_lottery.txt, docs/measurement.md) — and a stream never overlaps itself."""
    import torch
    from sfm_mvs_amd import ops
    dev = torch.device("cuda", 0)
    chain = ops.independent_streams(1, dev, priority=-1)[0]
    feat = ops.independent_streams(3, dev, avoid=[chain])
    ops.release_probe_scratch(dev)
    assert len(feat) == 3 and len({s.cuda_stream for s in feat + [chain]}) == 4
    assert not ops.streams_overlap(chain, chain)                 # one queue: strictly in order
    ok = sum(ops.streams_overlap(a, b) for a in feat + [chain] for b in feat + [chain] if a is not b)
    assert ok >= 10, ok                                          # 12 ordered pairs; the probe itself is a timing measurement: allow two misses
    ops.release_probe_scratch(dev)
