"""Dev tool: add the TunableOp entries of a second results file to the committed one (keys = operator signature + shape; entries already
present are kept as they are).  usage: merge_tuned.py profiles/tunableop_gfx950.csv new.csv"""
import sys
base, new = sys.argv[1:3]
lines = open(base).read().splitlines()
have = {",".join(l.split(",")[:2]) for l in lines if l and not l.startswith("Validator")}
add = [l for l in open(new).read().splitlines() if l and not l.startswith("Validator") and ",".join(l.split(",")[:2]) not in have]
open(base, "w").write("\n".join(lines + add) + "\n")
print(f"{len(add)} entries added to {base} ({len(have)} kept)")
for l in add:
    print("  +", l[:160])
