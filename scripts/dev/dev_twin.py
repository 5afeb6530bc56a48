import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from datagen import gustav_scene
from sfm_mvs_amd import pipeline as pl, ops
K,P,feats,ids=gustav_scene(9,seed=3,pix_noise=0.05)
b=pl.run_sfm(feats,K,device_resident=False, log=print)
print(len(b['errors']), b['Xtot'].shape)
for i in range(8):
    p0,p1=pl.match_features(feats[i],feats[i+1]); print(i,len(p0))
