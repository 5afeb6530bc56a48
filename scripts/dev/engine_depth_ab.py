"""Dev: same-box A/B of sharded.HipMatchEngine's pipeline depth on the all-pairs leg (32 images) and the headline step."""
import copy, os, sys, json, io, contextlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.argv = ["bench.py"]
import torch, bench
from sfm_mvs_amd import sharded
args = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
init = sharded.HipMatchEngine.__init__
for depth in (3, 2, 3, 2, 3, 2):
    sharded.HipMatchEngine.__init__.__defaults__ = (0.70, depth, 8, 4)
    a = copy.copy(args); a.images, a.verify_images, a.no_cpu_baseline = 32, 0, True
    r = bench.bench_allpairs(a, 1, 0, dev)
    a2 = copy.copy(args); a2.pipe_depth, a2.steps, a2.warmup, a2.no_extras, a2.no_cpu_baseline = depth, 60, 10, True, True
    k = bench.bench_knn(a2, 1, 0, dev)
    print(f"depth {depth}: allpairs {r['value']:.4g} /s   knn step {k['ms_per_step']:.4f} ms ({k['value']:.4g}/s)", flush=True)
