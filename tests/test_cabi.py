"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/mi_restore.h declares, the ctypes table matches the header, sizing entry points work without a GPU,
the modules keep the reference's state_dict, and CPU tensors are refused (no fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from image_restoration_amd import _lib
    return _lib


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "mi_restore.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(lib):
    handle = C.CDLL(lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 24
    for s in syms:
        assert hasattr(handle, s), f"{s} declared in include/mi_restore.h but not exported"
    assert set(syms) == set(lib.SIGNATURES), "ctypes table and header disagree"
    assert lib.lib().mi_version() == 100


def test_sizing_and_argument_errors_without_gpu(lib):
    L = lib
    s = L.MdtaShape(2, 48, 1, 16, 16, L.MI_F32, 3)
    assert L.lib().mi_mdta_saved_bytes(C.byref(s)) > 2 * 2 * 144 * 256 * 4
    assert L.lib().mi_mdta_workspace(C.byref(s)) > 0
    g = L.GdfnShape(2, 48, 127, 16, 16, L.MI_BF16, 3)
    # conv input (2h planes) + gate output (h planes); the conv output is recomputed in backward for 3x3 on 16-pixel rows
    assert L.lib().mi_gdfn_saved_bytes(C.byref(g)) >= 2 * (254 + 127) * 256 * 2
    bad = L.MdtaShape(2, 50, 4, 16, 16, L.MI_F32, 3)   # 50 channels not divisible by 4 heads
    assert L.lib().mi_mdta_saved_bytes(C.byref(bad)) == 0
    assert b"divisible" in L.lib().mi_last_error()
    # null pointers are rejected with an error code, not a crash
    assert L.lib().mi_ln_fwd(None, None, None, None, None, None, 1, 4, 16, 1, 0, None) == -1
    assert L.lib().mi_ln_bwd_workspace(8, 48, 65536) > 0
    assert L.lib().mi_dwconv_bwd_workspace(8, 144, 256, 256, 3) > 0


def test_state_dict_matches_reference_and_cpu_is_refused(lib):
    import image_restoration_amd as m
    from oracle import restormer_ref as R
    from oracle.fixtures import load
    net = m.Restormer()
    keys = [str(k) for k in load("restormer_base_keys")["keys"]]
    sd = net.state_dict()
    assert list(sd) == keys
    shapes = R.restormer_param_shapes(R.RESTORMER_BASE)
    assert {k: tuple(v.shape) for k, v in sd.items()} == shapes
    assert sum(p.numel() for p in net.parameters()) == 26126644
    tiny = m.Restormer(**{k: v for k, v in R.RESTORMER_TINY.items()})
    tiny.load_state_dict(R.make_restormer_state(R.RESTORMER_TINY, seed=1))
    blk = m.TransformerBlock(48, 1, 2.66, True, "BiasFree")
    assert "norm1.body.bias" not in blk.state_dict() and "attn.qkv.bias" in blk.state_dict()
    with pytest.raises(RuntimeError, match="MI355X only"):
        blk(torch.zeros(1, 48, 8, 8))
    with pytest.raises(RuntimeError, match="MI355X only"):
        m.LayerNorm(48, "WithBias")(torch.zeros(1, 48, 8, 8))


def test_moce_and_adair_state_dict_keys_match_reference(lib):
    """Key lists captured from the reference modules (tools/capture_golden_moce.py) vs the drop-in modules."""
    import image_restoration_amd.adair as ad
    import image_restoration_amd.moce_ir as mo
    from oracle.fixtures import load
    gold = load("moce_keys")
    db = mo.DecoderBlock(dim=48, num_heads=1, ffn_expansion_factor=2, bias=False, LayerNorm_type="WithBias",
                         expert_layer=mo.FFTAttention, complexity_scale="max", rank=2, num_experts=4, top_k=1,
                         depth_type="constant", rank_type="spread", stage_depth=1, freq_dim=64, with_complexity=True)
    assert list(db.state_dict()) == [str(k) for k in gold["decoder"]]
    assert list(mo.EncoderBlock(48, 2, 2, True, "WithBias").state_dict()) == [str(k) for k in gold["encoder"]]
    assert list(mo.CrossAttention(48, 1, True).state_dict()) == [str(k) for k in gold["cross"]]
    assert list(ad.Chanel_Cross_Attention(48, 4, False).state_dict()) == [str(k) for k in gold["adair_cross"]]
    with pytest.raises(RuntimeError, match="MI355X only"):
        mo.CrossAttention(48, 1, True)(torch.zeros(1, 48, 8, 8), torch.zeros(1, 48, 8, 8))
    # AdaIR: the whole network and its frequency module (tools/capture_golden_adair.py), MoCE-IR whole network
    from image_restoration_amd import configs
    from oracle import adair_ref as A
    akeys = load("adair_keys")
    net = ad.AdaIR(**configs.ADAIR_BASE)
    assert list(net.state_dict()) == [str(k) for k in akeys["adair_base"]]
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == A.adair_param_shapes(configs.ADAIR_BASE)
    assert list(ad.FreModule(48, 4, False).state_dict()) == [str(k) for k in akeys["fre"]]
    with pytest.raises(RuntimeError):
        ad.FreModule(32, 2, False)(torch.zeros(1, 3, 32, 32), torch.zeros(1, 32, 8, 8))      # CPU tensors: no fallback


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "image_restoration_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f"{f} imports the oracle"
