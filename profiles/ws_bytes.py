import sys; sys.path.insert(0,'/root/repo')
import torch, sapcu_amd
from sapcu_amd import _lib, testing as T
fn = sapcu_amd.ImprovedSNNNormalEstimation(k_values=[24,18,12], emb_dims=640, time_steps_enc=4, time_steps_dec=12, num_heads=8).cuda()
fd = sapcu_amd.EnhancedSNNDistanceEstimation(k=32, emb_dims=768, time_steps_enc=4, time_steps_dec=8, num_heads=8, dropout=0.1, use_snn_decoder=False, k_scales=[8,16,32,48]).cuda()
lib=_lib.load()
for m in (fn, fd):
    h = m._engine()
    for b, pts in ((4096, 48), (4096, 100), (256, 48)):
        print(type(m).__name__, b, pts, "%.2f GB" % (lib.sapcu_workspace_bytes(h, b, pts)/1e9))
