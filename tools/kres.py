"""Per-kernel resources (VGPRs, SGPRs, LDS bytes, scratch, spills) from hipcc -S output: python tools/kres.py file.s [substring]"""
import re, sys
txt = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    if sub in name:
        print(f"{name[:70]:70s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>4s} spill {g('vgpr_spill_count'):>3s}")
