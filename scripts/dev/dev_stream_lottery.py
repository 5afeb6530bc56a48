"""Dev: how much of the run-to-run spread of the pipelined legs is the stream -> hardware-queue lottery?
SIFT (3 frames in flight) and the KNN batch pipeline (2 launch sets in flight) on plain fresh streams vs ops.independent_streams."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from sfm_mvs_amd import ops, sift
from datagen import scene_image
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
junk = [torch.cuda.Stream(device=dev) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5)]     # a process that already made some streams
w, h = 968, 648
gray = torch.as_tensor(scene_image(w, h, 3)).to(dev)
def fps(pipe, n=60):
    for _ in range(6): pipe.submit(gray, after=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): pipe.submit(gray, after=False)
    torch.cuda.synchronize(); return n / (time.perf_counter() - t0)
def serial(ss):
    return [f"{a}->{b}" for a in range(len(ss)) for b in range(len(ss)) if a != b and not ops.streams_overlap(ss[a], ss[b])]
pipe = sift.SiftPipeline(w, h, dev, depth=3)
for trial in range(6):
    pipe.streams = [torch.cuda.Stream(device=dev) for _ in range(3)]; junk += pipe.streams
    f = fps(pipe)
    print(f"sift plain streams #{trial}: {f:.0f} frames/s  serialised {serial(pipe.streams)}", flush=True)
for trial in range(4):
    pipe.streams = ops.independent_streams(3, dev); junk += pipe.streams
    print(f"sift independent streams #{trial}: {fps(pipe):.0f} frames/s", flush=True)
# KNN: 8 pairs per launch set, 2 in flight
g = torch.Generator(device="cpu"); g.manual_seed(0)
sets = [[(torch.rand(10000, 128, generator=g).to(dev), torch.rand(10000, 128, generator=g).to(dev)) for _ in range(8)] for _ in range(2)]
bp = ops.BatchPipeline(10000, 10000, dev, batch=8, depth=2)
for trial in range(8):
    bp.streams = [torch.cuda.Stream(device=dev) for _ in range(2)]; junk += bp.streams
    ms = bp.probe_ms(sets, 60)
    print(f"knn plain streams #{trial}: {ms:.4f} ms per launch set  serialised {serial(bp.streams)}", flush=True)
for trial in range(4):
    bp.streams = ops.independent_streams(2, dev); junk += bp.streams
    print(f"knn independent streams #{trial}: {bp.probe_ms(sets, 60):.4f} ms per launch set", flush=True)
