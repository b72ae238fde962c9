"""kmeta.py file.s [filter]: registers / scratch / LDS of every kernel in an assembly listing (amdhsa metadata)."""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in re.split(r"\n  - \.agpr_count:", txt)[1:]:
    blk = ".agpr_count:" + blk
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    if flt in name:
        print(f"{name[:70]:70s} vgpr {g('vgpr_count'):>4} agpr {g('agpr_count'):>4} spill {g('vgpr_spill_count'):>3} scratch {g('private_segment_fixed_size'):>5} lds {g('group_segment_fixed_size'):>6}")
