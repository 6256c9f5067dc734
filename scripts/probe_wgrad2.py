import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mst_amd.model import MixingStyleEncoder, HipEncoder
B, Fr = 72, 1723
torch.manual_seed(0)
model = MixingStyleEncoder(44100, 1024, 256, 128, 20, 10, 8, 768, feature_dim=64).cuda().eval()
enc = HipEncoder(model, "fp32")
lm = torch.randn(B, 8, 128, Fr, device="cuda")
feats = torch.randn(B, 64, device="cuda")
with torch.no_grad():
    _, t = enc.forward_train(lm, feats=feats, head=False)
    p1 = t["pool1"]
    for _ in range(2): enc.conv2_wgrad(p1, B, Fr)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): enc.conv2_wgrad(p1, B, Fr)
    e1.record(); torch.cuda.synchronize()
print(f"MST_WGRAD_DBG={os.environ.get('MST_WGRAD_DBG','0')}: conv2_wgrad {e0.elapsed_time(e1)/3:.2f} ms")
# same kernel on a real gradient (backward_apply writes d(conv2 output) in accumulator order, values ~1e-7)
with torch.no_grad():
    for scale in (1.0, 1e-6):
        dfilm = torch.zeros(B, enc.n_sub * 192, device="cuda")
        dpool = torch.randn(B, 64 * enc.n_sub * enc.freq_dim, (Fr // 5) // 4, device="cuda") * scale
        enc.forward_train(lm, feats=feats, head=False)
        enc.backward_apply(2, dpool, dfilm, B, Fr)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3): enc.conv2_wgrad(p1, B, Fr)
        e1.record(); torch.cuda.synchronize()
        print(f"after backward_apply(2), dpool scale {scale}: conv2_wgrad {e0.elapsed_time(e1)/3:.2f} ms")
