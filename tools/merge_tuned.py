"""Merge freshly tuned hipBLASLt solution records into lemon_amd/data/linear_gfx950.csv: keys (m, n, k, epilogue, residual,
operand type) of the new file replace the recorded ones, everything else stays.  Both files must carry the same stamp line.
python tools/merge_tuned.py gpurun_out/linear_gfx950.csv [target.csv]"""
import os, sys

src = sys.argv[1]
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lemon_amd", "data", "linear_gfx950.csv")


def read(path):
    head, rows = [], {}
    for line in open(path):
        line = line.rstrip("\n")
        if not line:
            continue
        if line.startswith("#"):
            head.append(line)
            continue
        f = line.split(",")
        key = tuple(f[:5]) + ((f[7],) if len(f) > 7 else ("0",))
        rows[key] = line
    return head, rows


h_new, new = read(src)
h_old, old = read(dst)
assert h_new[0] == h_old[0], f"stamp mismatch: {h_new[0]!r} vs {h_old[0]!r}"
replaced = sum(1 for k in new if k in old)
old.update(new)
with open(dst, "w") as f:
    f.write("\n".join(h_old) + "\n")
    for k in sorted(old, key=lambda k: (int(k[5]), int(k[2]), int(k[1]), int(k[0]), int(k[3]), int(k[4]))):
        f.write(old[k] + "\n")
print(f"{len(new)} keys merged into {dst} ({replaced} replaced, {len(old)} total)")
