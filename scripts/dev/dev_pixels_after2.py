"""Dev: which part of the DEFAULT bench run slows the overlapped from-pixels job down (0.061 s alone, 0.076-0.087 s as the run's last leg)?"""
import copy, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.argv = ["bench.py"]
import torch, bench
args = bench.parse(); args.no_cpu_baseline = True; args.images = 57
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
import sfm_mvs_amd; sfm_mvs_amd.lib()
def px(tag):
    a = copy.copy(args); a.steps = 3; a.warmup = 1
    r = bench.bench_sfm_pixels(a, 1, 0, dev)
    import threading
    print(f"{tag}: from pixels {r['value']*1e3:.1f} ms   threads {threading.active_count()}  allocated {torch.cuda.memory_allocated() >> 20} MiB reserved {torch.cuda.memory_reserved() >> 20} MiB", flush=True)
px("fresh process")
bench.bench_knn(copy.copy(args), 1, 0, dev); torch.cuda.synchronize(); px("after the headline leg (bench_knn)")
bench.extras(dev); torch.cuda.synchronize(); px("after extras() (triangulation legs)")
bench.extra_c4(dev); torch.cuda.synchronize(); px("after extra_c4 (dense BA, 500 x 200k)")
for name, fn, over in (("config5", bench.bench_c5, {}), ("allpairs", bench.bench_allpairs, {"images": 32, "verify_images": 4}),
                       ("sift", bench.bench_sift, {"steps": 30, "warmup": 5}), ("sfm57", bench.bench_sfm, {"steps": 2, "warmup": 1})):
    a = copy.copy(args)
    for k, v in over.items(): setattr(a, k, v)
    fn(a, 1, 0, dev); torch.cuda.synchronize()
    px("after " + name)
torch.cuda.empty_cache(); px("after empty_cache()")
