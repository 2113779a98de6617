"""VGPRs / scratch / LDS of every kernel in a hipcc -save-temps .s file (the amdhsa metadata at its end)."""
import re, sys
txt = open(sys.argv[1] if len(sys.argv) > 1 else "build/emsar_hip-hip-amdgcn-amd-amdhsa-gfx950.s").read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    if pat in name:
        print("%-90s vgpr %s scratch %s lds %s sgpr %s" % (name[:90], g("vgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), g("sgpr_count")))
