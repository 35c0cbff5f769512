#!/usr/bin/env python
"""Rank the kernels of a `tools/pmc_any.sh <dir> "" -- <command>` run by LDS bank conflicts per LDS cycle, with the share of a wave's
life spent on the VALU, waiting for anything, waiting for LDS.  Own kernels only (transoar:: and the anonymous-namespace ones,
which rocprofv3 prints as "void").      python tools/pmc_rank.py gpurun_out/<tag>_pmc_all.txt"""
import ast
import re
import sys

rows = []
for ln in open(sys.argv[1]):
    m = re.match(r"(.*) waves (\d+) (\{.*\})$", ln.strip())
    if not m:
        continue
    name, waves, d = m.group(1), int(m.group(2)), ast.literal_eval(m.group(3))
    if "transoar" not in name and name.strip() != "void":
        continue
    wc = max(d.get("WAVE_CYCLES", 1), 1)
    rows.append((d.get("LDS_BANK_CONFLICT", 0) / max(d.get("LDS_IDX_ACTIVE", 1), 1), name[-56:], waves, d, wc))
rows.sort(key=lambda r: -r[0])
print("%-56s %9s %9s %9s %6s %9s %9s" % ("kernel", "waves", "conf/lds", "lds/wave", "valu", "wait_any", "wait_lds"))
for r, name, waves, d, wc in rows:
    print("%-56s %9d %9.2f %9.2f %6.2f %9.2f %9.2f" % (name, waves, r, d.get("LDS_IDX_ACTIVE", 0) / wc, d.get("ACTIVE_INST_VALU", 0) / wc,
                                                      d.get("WAIT_ANY", 0) / wc, d.get("WAIT_INST_LDS", 0) / wc))
