#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: python tools/pmc_summary.py <csv> [<csv> ...]"""
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        agg[k]["_n_" + r["Counter_Name"]] += 1
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if wc <= 0 or ("mi" not in k):
        continue
    waves = c.get("SQ_WAVES", 1)
    pct = lambda n: 100.0 * c.get(n, 0) / wc
    print(f"{k}\n   WAIT_ANY {pct('SQ_WAIT_ANY'):5.1f}%  WAIT_INST {pct('SQ_WAIT_INST_ANY'):5.1f}%  ACTIVE {pct('SQ_ACTIVE_INST_ANY'):5.1f}% "
          f"(VALU {pct('SQ_ACTIVE_INST_VALU'):4.1f}% LDS {pct('SQ_ACTIVE_INST_LDS'):4.1f}%)  waves {waves:.0f}")
    if "SQ_INSTS_VALU" in c:
        w2 = c.get("SQ_WAVES", waves)
        print(f"   insts/wave: VALU {c['SQ_INSTS_VALU']/w2:.0f} LDS {c.get('SQ_INSTS_LDS',0)/w2:.0f} VMEM_RD {c.get('SQ_INSTS_VMEM_RD',0)/w2:.0f} "
              f"VMEM_WR {c.get('SQ_INSTS_VMEM_WR',0)/w2:.0f} SALU {c.get('SQ_INSTS_SALU',0)/w2:.0f}  bank-conflict/LDS-active "
              f"{100*c.get('SQ_LDS_BANK_CONFLICT',0)/max(c.get('SQ_LDS_IDX_ACTIVE',1),1):.0f}%  WAIT_INST_LDS {100*c.get('SQ_WAIT_INST_LDS',0)/wc:.1f}%")
    if "SQ_INSTS_MFMA" in c:
        w3 = c.get("SQ_WAVES", waves)
        print(f"   MFMA/wave {c['SQ_INSTS_MFMA']/w3:.0f}  MFMA busy cycles / wave-quad-cycles {100*c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(4*wc):.1f}%  "
              f"TRANS/wave {c.get('SQ_INSTS_VALU_TRANS',0)/w3:.0f}  VMEM inst cycles {100*c.get('SQ_INST_CYCLES_VMEM',0)/wc:.1f}%")
