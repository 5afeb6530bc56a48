import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from oracle import oracle as O
from datagen import gustav_scene
from oracle_backend import oracle_pipeline_backend
from sfm_mvs_amd import pipeline as pl
K,P,feats,ids=gustav_scene(12,seed=7,pix_noise=0.2)
got=pl.run_sfm(feats,K); want=pl.run_sfm(feats,K,be=oracle_pipeline_backend(O))
g=np.array(got['errors']); w=np.array(want['errors'])
print('errors rel diff',np.abs(g-w)/w)
print('pose absdiff per cam',np.abs(got['posearr']-want['posearr'])[9:].reshape(-1,12).max(1))
print('X rel', (np.abs(got['Xtot']-want['Xtot'])/(np.abs(want['Xtot'])+1e-6)).max())
