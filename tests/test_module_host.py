"""Host-side behaviour of the drop-in ``L1TensorProduct`` (no GPU): constructor contract, state dict,
initial-weight RNG stream, norm buffers, masks, instructions and error behaviour — against what the
unmodified reference produced (tests/golden/l1tp_meta.json + *.npz)."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_case
from models.segnn.l1_tensor_prod import L1TensorProduct
from scalable_e3_gnn_amd import Irreps

META = json.load(open(os.path.join(GOLDEN, "l1tp_meta.json")))


def build(c):
    torch.manual_seed(c["seed"])
    return L1TensorProduct(Irreps(c["in1"]), Irreps(c["out"]) if c["out"] else None, **c["kwargs"])


@pytest.mark.parametrize("name", sorted(META["cases"]))
def test_ctor_matches_reference(name):
    c = META["cases"][name]
    z = load_case(name)
    mod = build(c)
    assert list(mod.state_dict().keys()) == c["state_dict_keys"]
    assert [k for k, _ in mod.named_parameters()] == c["param_names"]
    assert [k for k, _ in mod.named_buffers()] == c["buffer_names"]
    for k, v in mod.state_dict().items():
        ref = z["init_" + k]
        assert tuple(v.shape) == ref.shape and v.dtype == torch.float32
        assert np.array_equal(v.numpy(), ref), f"{k}: same seed must give the reference's initial value"
    for k, want in c["attrs"].items():
        assert getattr(mod, k) == want, k
    for k, want in c["masks"].items():
        m = getattr(mod, k)
        assert m.dtype == torch.bool and m.device.type == "cpu" and m.to(torch.int8).tolist() == want
    got = [[i.i_in1, i.i_in2, i.i_out, i.connection_mode, i.has_weight, i.path_weight, list(i.path_shape)]
           for i in mod.instructions]
    assert got == c["instructions"]
    assert str(mod.iri2) == "1x0e+1x1o" and mod.in2_dim == 4


def test_masks_are_not_buffers_and_stay_on_cpu():
    mod = L1TensorProduct(Irreps("8x0e+8x1o"))
    assert "iri1_l0e" not in dict(mod.named_buffers())
    assert [tuple(b.shape) for _, b in mod.named_buffers()] == [(8,), (0,), (0,), (24,)]


def test_norm_buffers_widen_rounded_fp32_value():  # Q4
    mod = L1TensorProduct(Irreps("2x0e+1x0o+2x1o+1x1e"), Irreps("2x0e+1x0o+1x1e+2x1o")).double()
    assert mod.norm_l0e[0].item() == 0.40824830532073975
    assert mod.norm_l1o[0].item() == 0.7745966911315918
    assert mod.instructions[0].path_weight != mod.norm_l0e[0].item()  # path_weight keeps the unrounded double


def test_accepts_strings_and_duck_typed_irreps():
    class Ir:
        def __init__(s, l, p): s.l, s.p, s.dim = l, p, 2 * l + 1
    class MulIr:
        def __init__(s, mul, ir): s.mul, s.ir, s.dim = mul, ir, mul * ir.dim
    class FakeE3nnIrreps(list):
        @property
        def lmax(s): return max(m.ir.l for m in s)
        @property
        def dim(s): return sum(m.dim for m in s)
    e3 = FakeE3nnIrreps([MulIr(4, Ir(0, 1)), MulIr(4, Ir(1, -1))])
    torch.manual_seed(3)
    a = L1TensorProduct(e3)
    torch.manual_seed(3)
    b = L1TensorProduct("4x0e+4x1o")
    assert a.iri1 is e3 and a.iro is e3
    assert all(torch.equal(x, y) for x, y in zip(a.state_dict().values(), b.state_dict().values()))


@pytest.mark.parametrize("name", sorted(META["errors"]))
def test_error_behaviour(name):
    rec = META["errors"][name]
    spec = rec["spec"]
    exc = {"AssertionError": AssertionError, "AttributeError": AttributeError, "IndexError": IndexError,
           "ZeroDivisionError": ZeroDivisionError}[rec["raised"]]
    with pytest.raises(exc) as ei:
        torch.manual_seed(0)
        mod = L1TensorProduct(Irreps(spec["in1"]), Irreps(spec["out"]) if spec.get("out") else None,
                              **spec.get("kwargs", {}))
        if spec.get("forward"):
            D1 = mod.in1_dim + spec.get("in1_dim_delta", 0)
            x = torch.zeros(2, 3, D1) if spec.get("in1_3d") else torch.zeros(3, D1)
            mod(x, torch.zeros(3, spec.get("in2_dim", 4)))
    if rec["raised"] in ("AssertionError", "AttributeError") and name != "in1_3d":
        assert str(ei.value) == rec["message"]


def test_cpu_tensors_fail_loudly_no_fallback():
    mod = L1TensorProduct(Irreps("4x0e+4x1o"))
    with pytest.raises(RuntimeError, match="no CPU path"):
        mod(torch.zeros(3, 16), torch.zeros(3, 4))


def test_state_dict_roundtrip_and_deepcopy():
    c = META["cases"]["mixed"]
    z = load_case("mixed")
    mod = build(c).double()
    sd = {k[3:]: torch.tensor(z[k]) for k in z.files if k.startswith("sd_")}
    mod.load_state_dict(sd, strict=True)
    mod2 = copy.deepcopy(mod)
    assert all(torch.equal(a, b) for a, b in zip(mod.state_dict().values(), mod2.state_dict().values()))


def test_modules_deepcopy_and_pickle_without_sharing_plan_handles():
    """plan handles are ctypes pointers; modules must still deep-copy / pickle (EMA copies, torch.save of a
    whole module) and two copies must never own the same handle."""
    import copy
    import pickle

    from scalable_e3_gnn_amd.segnn import SEGNN

    m = SEGNN("1x0e+1x1o", 32, "1x1o", 2, lmax=2)
    m.layers[0].msg1._plan.handle(None)            # a host-side handle exists
    c = copy.deepcopy(m)
    assert c.layers[0].msg1._plan is not m.layers[0].msg1._plan
    assert c.layers[0].msg1._plan._plans._handles == {} or \
        c.layers[0].msg1._plan._plans._handles[-1].value != m.layers[0].msg1._plan._plans._handles[-1].value
    assert c.layers[0]._msg is not m.layers[0]._msg
    p = pickle.loads(pickle.dumps(m))
    assert sorted(p.state_dict()) == sorted(m.state_dict())
    l1 = SEGNN("1x0e+1x1o", 32, "1x1o", 1, lmax=1)
    l1.layers[0].msg1._get_plan().handle(None)
    c1 = copy.deepcopy(l1)
    assert c1.layers[0].msg1._plan is not l1.layers[0].msg1._plan


def test_fused_availability_does_not_depend_on_first_call_grad_mode():
    """the cached part of the fused-path decision is static; grad mode is looked at on every call."""
    import torch

    from scalable_e3_gnn_amd.segnn import SEGNNLayer

    layer = SEGNNLayer(32, 2)
    with torch.enable_grad():
        a = layer.fused_available()
    with torch.no_grad():
        b = layer.fused_available()
    assert a == b == SEGNNLayer(32, 2).fused_available()
