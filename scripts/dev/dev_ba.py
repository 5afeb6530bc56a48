import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from datagen import ba_problem
from sfm_mvs_amd import ba
for perturb in (0.003, 0.01, 0.03):
    K, cams, X, obs = ba_problem(6, 800, 0.3, seed=8, perturb=perturb)
    c, x, h = ba.bundle_adjust(torch.from_numpy(cams).cuda(), K, torch.from_numpy(X).cuda(), torch.from_numpy(obs).cuda(), iters=30)
    print(perturb, "noise floor", 2*6*800*0.09, [round(v,1) for v in h[:4]], "...", [round(v,1) for v in h[-3:]], len(h))
