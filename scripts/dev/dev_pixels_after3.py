"""Dev: the from-pixels job called for the FIRST time after all other legs of the default run (as the run does), then again."""
import copy, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.argv = ["bench.py"]
import torch, bench
args = bench.parse(); args.no_cpu_baseline = True; args.images = 57
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
import sfm_mvs_amd; sfm_mvs_amd.lib()
def px(tag, warm=1, steps=3):
    a = copy.copy(args); a.steps = steps; a.warmup = warm
    r = bench.bench_sfm_pixels(a, 1, 0, dev)
    print(f"{tag}: from pixels {r['value']*1e3:.1f} ms  reserved {torch.cuda.memory_reserved() >> 20} MiB", flush=True)
bench.bench_knn(copy.copy(args), 1, 0, dev); bench.extras(dev); bench.extra_c4(dev)
for name, fn, over in (("config5", bench.bench_c5, {}), ("allpairs", bench.bench_allpairs, {"images": 32, "verify_images": 4}),
                       ("sift", bench.bench_sift, {"steps": 30, "warmup": 5}), ("sfm57", bench.bench_sfm, {"steps": 2, "warmup": 1})):
    a = copy.copy(args)
    for k, v in over.items(): setattr(a, k, v)
    fn(a, 1, 0, dev); torch.cuda.synchronize()
px("first call, after every other leg (warmup 1)")
px("second call (warmup 1)")
from sfm_mvs_amd import pipeline as pl
for k in range(6):
    pl._SIFT_PIPES.clear()                      # new SIFT pipelines, new streams, a new chain stream
    px(f"fresh streams #{k}")
