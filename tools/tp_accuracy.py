import sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import models
from oracle import tp_oracle as T
from scalable_e3_gnn_amd.tensor_product import SHTensorProduct
torch.manual_seed(0)
in1, out = "32x0e+32x1o+32x2e+32x0e+32x1o+32x2e+1x0e", "96x0e+32x1o+32x2e"
mod = SHTensorProduct(in1, out, 2).to("cuda:0")
B = 4096
g = torch.Generator().manual_seed(1)
x = torch.randn(B, mod.in1_dim, generator=g, dtype=torch.float64)
y = torch.randn(B, mod.in2_dim, generator=g, dtype=torch.float64)
W = {c: getattr(mod, "weights_" + c).detach().double().cpu().numpy() for c in T.CLASSES if hasattr(mod, "weights_" + c)}
N = {c: getattr(mod, "norm_" + c).double().cpu().numpy() for c in T.CLASSES}
want = T.forward(in1, out, 2, x.float().double().numpy(), y.float().double().numpy(), W, N)
with torch.no_grad():
    got = mod.forward_fused([(x.float().to("cuda:0"), None)], y.float().to("cuda:0"), gate=False).double().cpu().numpy()
d = got - want
print("max rel (max|d|/max|want|) %.3e   rms rel %.3e" % (np.abs(d).max() / np.abs(want).max(), np.sqrt((d**2).mean()) / np.sqrt((want**2).mean())))
