"""Register / spill / LDS table of every kernel in a hipcc -save-temps assembly file.  Usage: python3 tools/kernel_regs.py <file.s> [name filter]"""
import re
import sys


def main():
    txt = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for blk in txt.split("  - .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
        name = g("name")
        if flt in name:
            print(f"{name[:90]:90s} vgpr {g('vgpr_count'):>4} spill {g('vgpr_spill_count'):>3} sgpr {g('sgpr_count'):>4} spill {g('sgpr_spill_count'):>3}")


if __name__ == "__main__":
    main()
