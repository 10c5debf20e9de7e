"""Rewrites the C3-strong table of DESIGN.md section 6 from profiles/r06_scaling_prediction.txt."""
import os, re
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
txt = open(os.path.join(R, "profiles", "r06_scaling_prediction.txt")).read()
blocks = [b for b in txt.split("\n\n") if b.lstrip().startswith("# C3")]
rows = {}
for bi, b in enumerate(blocks[:2]):
    for line in b.splitlines():
        if line.startswith("#") or not line.strip():
            continue
        m = re.match(r"\s*(\d+)\s+(\d+)\s+([\d.]+)\s+(.*)$", line)
        G, Mr, ms, rest = int(m.group(1)), int(m.group(2)), m.group(3), m.group(4)
        rest = re.sub(r"\s+", " ", rest.strip())
        if G == 1:
            cell = f"{ms} ms — {rest.split()[0]} evals/s"
        else:
            mm = re.match(r"([\d.e+]+) - ([\d.e+]+) ([\d.]+) - ([\d.]+) \(no xGMI allowance: ([\d.]+)\)", rest)
            cell = f"{ms} ms — {mm.group(1)}–{mm.group(2)} evals/s — eff. {mm.group(3)}–{mm.group(4)} ({mm.group(5)} without the allowance)"
        rows.setdefault(G, [Mr, None, None])[1 + bi] = cell
table = ["| G | M per rank | torch.distributed nccl binding (`bench.py`'s default) | library communicator (`--collective capi`) |", "|---|---|---|---|"]
for G in sorted(rows):
    table.append(f"| {G} | {rows[G][0]} | {rows[G][1]} | {rows[G][2]} |")
p = os.path.join(R, "DESIGN.md")
s = open(p).read()
i = s.index("| G | M per rank | torch.distributed nccl binding")
j = s.index("\n\n", i)
s = s[:i] + "\n".join(table) + s[j:]
open(p, "w").write(s)
print("\n".join(table))
