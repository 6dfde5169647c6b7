"""The kernel-selection knobs are read once per process (E3_TP_AB, E3_TP_EXACT, E3_FUSED_SCATTER), so each mode is
exercised in a child process: fused l_max=2 message products + one SEGNN layer against the fp64 oracle chain."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, %(repo)r)
import models  # noqa: F401
from oracle import segnn_oracle as S
from scalable_e3_gnn_amd import ops
from scalable_e3_gnn_amd.radius_graph import radius_graph
from scalable_e3_gnn_amd.segnn import SEGNN
torch.manual_seed(8)
N, H, L = 300, 32, 1
pos = torch.rand(N, 3, generator=torch.Generator().manual_seed(8))
r = float((3 * 10.0 / (4 * np.pi * N)) ** (1 / 3))
model = SEGNN("1x0e+1x1o", H, "1x1o", L, lmax=2).to("cuda:0")
g = radius_graph(pos.to("cuda:0"), r, [0, 0, 0], [1, 1, 1])
xs = torch.randn(N, 4, generator=torch.Generator().manual_seed(9))[g.perm.cpu().long()]
with torch.no_grad():
    assert model.layers[0]._fused()
    out = model(xs.to("cuda:0"), g).double().cpu().numpy()
params = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
want = S.forward_l2(params, H, L, "1x0e+1x1o", "1x1o", xs.double().numpy(), pos.numpy()[g.perm.cpu().numpy()],
                    g.rowptr.cpu().numpy(), g.src.cpu().numpy())
err = float(np.abs(out - want).max() / np.abs(want).max())
print("ERR", err)
assert err < 1e-4, err
"""


@pytest.mark.gpu
@pytest.mark.parametrize("env", [
    {"E3_TP_R16": "0", "E3_TP_AB": "0"},                      # one wave per 32-row tile (e3_tp_mfma.hip)
    {"E3_TP_R16": "0", "E3_TP_AB": "0", "E3_TP_EXACT": "1"},  # exact fp32 MFMA operands
    {"E3_TP_R16": "0"},                                       # two waves per 32-row tile (e3_tp_mfma_ab.hip)
    {"E3_FUSED_SCATTER": "0"},                                # default kernel, separate (reproducible) segment-sum
    {"E3_TP_R16": "0", "E3_TP_AB": "0", "E3_TP_NBUF": "2"},   # double-buffered staging
])
def test_kernel_selection_modes(env):
    e = dict(os.environ)
    e.update(env)
    p = subprocess.run([sys.executable, "-c", CHILD % {"repo": REPO}], env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "ERR" in p.stdout
