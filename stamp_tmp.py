import os, sys, numpy as np
sys.path.insert(0, '.')
import torch
import cpprob_amd as cp
from cpprob_amd import capi
capi.LIB_PATH = os.path.join('cpprob_amd', 'lib', 'libcpprob_hip_stamps.so')
z = np.load('tests/golden/observations.npz')
eng = cp.Engine(0)
for n in (1000000, 4000000):
    eng.begin(cp.ALG_SMC, cp.MODEL_HMM3, z['hmm16'], n, seed=1, ess_threshold=2.0)
    for i in range(5): eng.run(i)
    eng.sync()
    os.environ['CPPROB_STAMP_DUMP'] = '1'
    eng.run(9); eng.sync()
    del os.environ['CPPROB_STAMP_DUMP']
