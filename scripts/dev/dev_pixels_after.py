"""Dev: does an earlier leg of the default bench run slow the overlapped from-pixels job down?"""
import copy, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.argv = ["bench.py"]
import torch, bench
args = bench.parse(); args.no_cpu_baseline = True; args.steps = 3; args.images = 57
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
def px(tag):
    r = bench.bench_sfm_pixels(copy.copy(args), 1, 0, dev)
    print(f"{tag}: from pixels {r['value']*1e3:.1f} ms   affinity {len(os.sched_getaffinity(0))} cpus", flush=True)
px("fresh process")
for name, fn, over in (("sfm57", bench.bench_sfm, {}), ("sift", bench.bench_sift, {"steps": 30, "warmup": 5}), ("allpairs", bench.bench_allpairs, {"images": 32, "verify_images": 4}),
                       ("config5", bench.bench_c5, {})):
    a = copy.copy(args)
    for k, v in over.items(): setattr(a, k, v)
    fn(a, 1, 0, dev); torch.cuda.synchronize()
    px("after " + name)
