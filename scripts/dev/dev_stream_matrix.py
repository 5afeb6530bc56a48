"""Dev: which stream pairs must be independent for the overlapped from-pixels job?  Late in a process (after the other bench legs)
create the job's four streams the plain way several times, time the job, and print the overlap matrix (ops.streams_overlap)."""
import copy, os, sys, time, itertools
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.argv = ["bench.py"]
import torch, bench
args = bench.parse(); args.no_cpu_baseline = True; args.images = 57
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
import sfm_mvs_amd; sfm_mvs_amd.lib()
from sfm_mvs_amd import ops, pipeline as pl
def px():
    a = copy.copy(args); a.steps = 3; a.warmup = 1
    return bench.bench_sfm_pixels(a, 1, 0, dev)["value"] * 1e3
bench.bench_knn(copy.copy(args), 1, 0, dev); bench.extras(dev)
for name, fn, over in (("config5", bench.bench_c5, {}), ("sift", bench.bench_sift, {"steps": 30, "warmup": 5})):
    a = copy.copy(args)
    for k, v in over.items(): setattr(a, k, v)
    fn(a, 1, 0, dev); torch.cuda.synchronize()
keep = []
for trial in range(10):
    pl._SIFT_PIPES.clear()
    chain = torch.cuda.Stream(device=dev, priority=-1)
    feat = [torch.cuda.Stream(device=dev) for _ in range(3)]
    keep += [chain] + feat
    pl._SIFT_PIPES[("job streams", 0, 3)] = (feat, chain)
    ms = px()
    names = ["chain", "f0", "f1", "f2", "default"]
    ss = [chain] + feat + [torch.cuda.default_stream(dev)]
    m = {(a, b): ops.streams_overlap(ss[a], ss[b]) for a in range(5) for b in range(5) if a != b}
    serial = [f"{names[a]}->{names[b]}" for (a, b), ok in m.items() if not ok]
    print(f"trial {trial}: from pixels {ms:.1f} ms   serialised pairs (long on first, short on second): {serial}", flush=True)
