import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mst_amd.model import MixingStyleEncoder
B, Fr = 72, 1723
torch.manual_seed(0)
model = MixingStyleEncoder(44100, 1024, 256, 128, 20, 10, 8, 768, feature_dim=64).cuda().eval()
from mst_amd.model import HipEncoder
enc = HipEncoder(model, "fp32")
lm = torch.randn(B, 8, 128, Fr, device="cuda")
feats = torch.randn(B, 64, device="cuda")
with torch.no_grad():
    enc.forward_train(lm, feats=feats, head=False)
    for _ in range(2): enc.conv1_wgrad(lm, B, Fr)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): enc.conv1_wgrad(lm, B, Fr)
    e1.record(); torch.cuda.synchronize()
print(f"MST_WGRAD_DBG={os.environ.get('MST_WGRAD_DBG','0')}: conv1_wgrad {e0.elapsed_time(e1)/5:.2f} ms")
