#!/usr/bin/env python3
"""Markdown table of per-kernel roofline evidence from a tools/pmc.sh summary (rocprofv3 --pmc passes, one counter set per pass).

    python3 tools/roofline_table.py gpurun_out/pmc_<tag>/<tag>_pmc_summary.json [min_us]

Derived columns (MI355X_MICROARCH.md, HBM / rocprofv3 section):
  VALU busy   = SQ_ACTIVE_INST_VALU x 4 / (kernel time x 2.1 GHz x 1024 SIMDs): the share of SIMD issue cycles spent on vector ALU
                instructions (SQ_ACTIVE_INST_* count in units of four cycles, summed over the chip; 2.1 GHz is what the part holds
                under these loads).  Equivalently (resident waves per SIMD) x (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES).
  waves/SIMD  = SQ_WAVE_CYCLES x 4 / (kernel time x 2.1 GHz x 1024)
  HBM read    = FETCH_SIZE KiB x 2 (gfx950 reports half of wide coalesced reads) / time;  HBM write = WRITE_SIZE KiB / time
  L2 hit      = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
"""
import json
import sys

d = json.load(open(sys.argv[1]))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
CLK, SIMDS = 2.1e9, 1024
print("| kernel | dispatches | avg us | VALU instr / wave | waves / SIMD | VALU busy | HBM read TB/s | HBM write TB/s | L2 hit |")
print("|---|---|---|---|---|---|---|---|---|")
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("_avg_us_profiled", 0) * kv[1].get("_dispatches", 0)):
    us = v.get("_avg_us_profiled", 0)
    if us < min_us:
        continue
    t = us * 1e-6
    waves = v.get("SQ_WAVES", 0)
    hit, miss = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
    print("| `%s` | %d | %.0f | %.0f | %.2f | %.0f %% | %.2f | %.2f | %.0f %% |" % (
        k.replace("moai::", ""), v.get("_dispatches", 0), us, v.get("SQ_INSTS_VALU", 0) / waves if waves else 0,
        v.get("SQ_WAVE_CYCLES", 0) * 4 / (t * CLK * SIMDS), 100 * v.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (t * CLK * SIMDS),
        v.get("FETCH_SIZE", 0) * 1024 * 2 / t / 1e12, v.get("WRITE_SIZE", 0) * 1024 / t / 1e12, 100 * hit / (hit + miss) if hit + miss else 0))
