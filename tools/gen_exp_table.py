"""Generates gingr_amd/csrc/exp_table*.inc: T[j] = 2^(j/N), correctly rounded float64 (mpmath, 200 bits), and prints the
polynomial constants of fastexp.h for that table size."""
import re
import sys

if len(sys.argv) > 1 and sys.argv[1] == "floor":
    # gingr_amd/csrc/exp_floor_table.inc: the table of the floor form, T'[j] = T[j] * S in float64 (round to nearest) from the
    # committed exp_table.inc -- what cpd_colsum / cpd_rowstats used to multiply out per workgroup
    vals = [float.fromhex(h) for h in re.findall(r"0x1\.[0-9a-f]+p\+0", open("gingr_amd/csrc/exp_table.inc").read())]
    S = float.fromhex("0x1.000000000038dp+0")
    with open("gingr_amd/csrc/exp_floor_table.inc", "w") as f:
        f.write("// T'[j] = fl(2^(j/2048)) * S, S = 0x1.000000000038dp+0 (fastexp.h: ExpFloor11::S), the product rounded to nearest float64 -- bit for\n"
                "// bit what the kernels used to compute per workgroup from exp_table.inc (tools/gen_exp_table.py floor)\n")
        for i in range(0, len(vals), 4):
            f.write("    " + ", ".join((v * S).hex() for v in vals[i:i + 4]) + ",\n")
    sys.exit(0)

from mpmath import mp, mpf, log

mp.prec = 200
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
out = sys.argv[2] if len(sys.argv) > 2 else "gingr_amd/csrc/exp_table.inc"
rows = []
for j in range(N):
    v = float(mpf(2) ** (mpf(j) / N))          # mpf -> float rounds to nearest
    rows.append(v.hex())
with open(out, "w") as f:
    f.write(f"// 2^(j/{N}), j = 0..{N - 1}, correctly rounded float64 (generated with mpmath at 200 bits)\n")
    for i in range(0, N, 4):
        f.write("    " + ", ".join(rows[i:i + 4]) + ",\n")
a = log(2) / N
h = mpf("0.5")
print("C1 ", repr(float(a)))
print("C2 ", repr(float(a * a / 2)))
print("C3 ", repr(float(a ** 3 / 6)))
print("C1_D2 (economised) ", repr(float(a + a ** 3 / 6 * mpf(3) / 4 * h * h)))
print("deg-2 economised error ", float(a ** 3 / 6 * h ** 3 / 4), " deg-3 Taylor remainder ", float((a * h) ** 4 / 24))
