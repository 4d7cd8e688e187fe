import sys, os, ctypes as C
sys.path.insert(0, 'nubomedia-vca_amd')
import numpy as np, torch
from nubovca import capi, synth
ctx = capi.Context(0)
# batch of 32 frames through the face path is what matters; emulate with the equalize+integral part only:
img = synth.make_gray(1920, 1080, 1, 'natural')
ctx.enable_kernel_timing(True)
for _ in range(20):
    ctx.integral(img)
print({k: round(v[0]/v[1]*1000, 1) for k, v in ctx.kernel_timing().items()}, 'us per launch (1 frame)')
