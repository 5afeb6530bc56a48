"""Dev: the from-pixels job in a FRESH process: job time and the overlap matrix of the streams pipeline._job_streams chose."""
import copy, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
mode = sys.argv[1] if len(sys.argv) > 1 else "probed"
sys.argv = ["bench.py"]
import torch, bench
args = bench.parse(); args.no_cpu_baseline = True; args.images = 57; args.steps = 4; args.warmup = 1
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
import sfm_mvs_amd; sfm_mvs_amd.lib()
from sfm_mvs_amd import ops, pipeline as pl
if mode == "plain":      # the order of the round before: three feature streams, then the chain stream
    feat = [torch.cuda.Stream(device=dev) for _ in range(3)]
    chain = torch.cuda.Stream(device=dev, priority=-1)
    pl._SIFT_PIPES[("job streams", 0, 3)] = (feat, chain)
ms = bench.bench_sfm_pixels(copy.copy(args), 1, 0, dev)["value"] * 1e3
feat, chain = pl._SIFT_PIPES[("job streams", 0, 3)]
names = ["chain", "f0", "f1", "f2", "default"]; ss = [chain] + list(feat) + [torch.cuda.default_stream(dev)]
serial = [f"{names[a]}->{names[b]}" for a in range(5) for b in range(5) if a != b and not ops.streams_overlap(ss[a], ss[b])]
print(f"{mode}: from pixels {ms:.1f} ms   serialised {serial}   rejected candidates {len(ops._REJECTED_STREAMS)}", flush=True)
