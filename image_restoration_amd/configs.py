"""Network configurations of the reference's three drop-in targets (constructor keyword dictionaries).

RESTORMER_BASE   Restormer.py:194-205 constructor defaults (26.13 M parameters): BASELINE.json configs[1], [2]
RESTORMER_TINY   the survey's pin of "Restormer-tiny" (configs[0]): dim 16, two blocks per level
MOCEIR_BASE      MoCE-IR-main/src/options.py:70-84 (`--model MoCE_IR`) as PLTrainModel passes them
                 (MoCE-IR-main/src/train.py:33-47; 25.35 M parameters): configs[3]
MOCEIR_S         options.py:55-68 (`--model MoCE_IR_S`, dim 32)
"""
RESTORMER_BASE = dict(dim=48, num_blocks=[4, 6, 6, 8], num_refinement_blocks=4, heads=[1, 2, 4, 8], ffn_expansion_factor=2.66,
                      bias=False, LayerNorm_type="WithBias", inp_channels=3, out_channels=3)
RESTORMER_TINY = dict(RESTORMER_BASE, dim=16, num_blocks=[2, 2, 2, 2], num_refinement_blocks=2)

MOCEIR_BASE = dict(dim=48, num_blocks=[4, 6, 6, 8], num_dec_blocks=[2, 4, 4], levels=4, heads=[1, 2, 4, 8],
                   num_refinement_blocks=4, topk=1, num_experts=4, rank=2, with_complexity=False, depth_type="constant",
                   stage_depth=[1, 1, 1], rank_type="spread", complexity_scale="max")
MOCEIR_S = dict(MOCEIR_BASE, dim=32)
# the tiny widths of the whole-network golden (tools/capture_golden_moce2.py): tests and capture rehearsals
MOCEIR_TINY = dict(dim=16, levels=4, heads=[1, 2, 4, 8], num_blocks=[1, 1, 1, 2], num_dec_blocks=[1, 1, 1],
                   num_refinement_blocks=1, rank=2, num_experts=4, depth_type="constant", stage_depth=[1, 1, 1],
                   rank_type="spread", topk=1, with_complexity=True, complexity_scale="max")
# AdaIR-main/net/model.py:380-390 constructor defaults (the Restormer U-Net + three FreModules; AdaIR-main/train.py builds it bare)
ADAIR_BASE = dict(dim=48, num_blocks=[4, 6, 6, 8], num_refinement_blocks=4, heads=[1, 2, 4, 8], ffn_expansion_factor=2.66,
                  bias=False, LayerNorm_type="WithBias", decoder=True)

# analytic work of one Restormer-base forward at 256 x 256: used by bench.py's whole-step roofline
RESTORMER_BASE_FWD_FLOP_PER_PIXEL = 4.80e6          # 314.5 GFLOP per 256^2 image; a training step is ~3x
RESTORMER_BASE_FWD_FUSED_BYTES_PER_PIXEL_BF16 = 755e6 / 65536.0   # 4 C N s per block over the 44 blocks
RESTORMER_BASE_MDTA_FWD_FLOP_PER_PIXEL = 27.78e9 / 65536.0        # q k^T + attn v over the 44 blocks
