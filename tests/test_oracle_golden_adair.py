"""Pins oracle/adair_ref.py against golden vectors captured from the imported reference AdaIR-main/net/model.py
(tools/capture_golden_adair.py): FreModule forward + every gradient with an empty and a non-empty low-frequency rectangle,
the assembled AdaIR network (tiny widths) forward, loss and parameter-gradient norms, the key lists.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import adair_ref as A
from oracle import restormer_ref as R
from oracle.fixtures import check, load, seeded_input

F64 = torch.float64
TINY = dict(dim=16, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1, heads=[1, 2, 2, 2], ffn_expansion_factor=2.66,
            bias=False, LayerNorm_type="WithBias", decoder=True)


@pytest.mark.parametrize("tag,dim,heads,img_hw,hw,B", [("c32", 32, 2, 64, 16, 2), ("c16_big", 16, 2, 768, 384, 2)])
def test_fre_module_vs_reference(tag, dim, heads, img_hw, hw, B):
    torch.set_num_threads(8)
    gold = load(f"adair_fre_{tag}")
    sd = {k: v.requires_grad_(True) for k, v in R.make_state(A.fre_param_shapes(dim, heads), 900 + dim, F64).items()}
    img = seeded_input((B, 3, img_hw, img_hw), 910 + dim, F64)
    y = seeded_input((B, dim, hw, hw), 911 + dim, F64).requires_grad_(True)
    with torch.no_grad():
        feat = torch.nn.functional.conv2d(torch.nn.functional.interpolate(img, (hw, hw), mode="bilinear"), sd["conv1.weight"], padding=1)
        assert np.array_equal(A.mask_half_sizes(feat, sd).numpy(), gold["half"])
    out = A.fre_module(img, y, sd, heads)
    out.backward(seeded_input(tuple(out.shape), 912 + dim, F64))
    check("y", out, gold, 1e-9)
    check("dy", y.grad, gold, 1e-9)
    for k, v in sd.items():
        if v.grad is not None and f"g_{k}.sub" in gold:
            check("g_" + k, v.grad, gold, 1e-9)
    # parameters the reference's forward never touches (conv, score_gen) or whose path is not differentiable (rate_conv)
    assert all(sd[k].grad is None or float(sd[k].grad.abs().max()) == 0.0 for k in ("conv.weight", "score_gen.weight", "rate_conv.0.weight"))


def test_adair_tiny_network_vs_reference():
    torch.set_num_threads(8)
    gold = load("adair_tiny_train")
    sd = {k: v.requires_grad_(True) for k, v in R.make_state(A.adair_param_shapes(TINY), 950, F64).items()}
    img = torch.rand((1, 3, 64, 64), generator=torch.Generator().manual_seed(951), dtype=F64)
    tgt = torch.rand((1, 3, 64, 64), generator=torch.Generator().manual_seed(952), dtype=F64)
    out = A.adair_forward(img, sd, TINY)
    loss = (out - tgt).abs().mean()
    loss.backward()
    check("y", out, gold, 1e-9)
    assert abs(float(loss) - float(gold["loss"])) < 1e-10
    names, norms = list(gold["grad_names"]), gold["grad_norms"]
    for k, n in zip(names, norms):
        k = str(k)
        got = float(sd[k].grad.norm()) if sd[k].grad is not None else -1.0
        if n < 0:
            assert got <= 0.0, k
        else:
            assert abs(got - n) <= 1e-8 * max(1.0, n), (k, got, n)


def test_adair_key_lists():
    keys = load("adair_keys")
    base = dict(dim=48, num_blocks=[4, 6, 6, 8], num_refinement_blocks=4, heads=[1, 2, 4, 8], ffn_expansion_factor=2.66,
                bias=False, LayerNorm_type="WithBias", decoder=True)
    assert list(A.adair_param_shapes(base)) == [str(k) for k in keys["adair_base"]]
    assert list(A.fre_param_shapes(48, 4)) == [str(k) for k in keys["fre"]]
