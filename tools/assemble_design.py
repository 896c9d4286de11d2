#!/usr/bin/env python3
"""One-off (round 3): DESIGN.md re-cut into "current state" sections plus appendices, from slices of the round-2 text
(/tmp/design_*.md, made with sed from `git show <round-2 commit>:DESIGN.md`) and the new text below.  Kept for the record of
what moved where; not part of the product or the tests."""
import sys
T = "/tmp/"
s1 = open(T + "design_s1.md").read()
s2 = open(T + "design_s2.md").read()
s3 = open(T + "design_s3.md").read()
s41 = open(T + "design_s41.md").read()
numerics = open(T + "design_numerics.md").read()
host = open(T + "design_host.md").read()
scope = open(T + "design_scope.md").read()
hist = open(T + "design_hist.md").read()
NEW = open(sys.argv[1]).read()
parts = {}
cur = None
for line in NEW.splitlines(keepends=True):
    if line.startswith("@@@ "):
        cur = line[4:].strip(); parts[cur] = ""
    elif cur:
        parts[cur] += line

s1 = s1.replace(parts["row_e_old"].strip("\n"), parts["row_e_new"].strip("\n"))
s1 = s1.replace(parts["row_f_old"].strip("\n"), parts["row_f_new"].strip("\n"))
s1 = s1.replace("Scope contract: `SURVEY.md` §8.  This file states what was built against each row of it.", parts["intro"].strip("\n"))
assert parts["row_e_new"].strip("\n") in s1 and parts["row_f_new"].strip("\n") in s1 and "Appendix A" in s1
s2 = s2.rstrip() + "\n" + parts["oracle_add"]
host = host.replace("### 4.5 Host planner: the reference's push order, speculation, long ribbon lists",
                    "## 5. Host planner: the reference's push order, speculation, long ribbon lists, the deadline")
host = host.replace("§4.2", "Appendix C").replace("§4.5", "§5").replace("§6 \"next\"", "§7")
host = host.rstrip() + "\n\n" + parts["deadline"]
appA = parts["appA_head"] + hist.split("|---|---|---|---|\n", 1)[1]
out = (s1 + "\n" + s2 + "\n" + s3 + "\n## 4. Kernels\n\n" + s41 + "\n" + parts["kern_table"] + parts["kernels_desc"] + parts["roof"] + "\n" + host + "\n" +
       parts["multi"] + parts["meas"] + "\n## 8. Out of scope (SURVEY §2 rows 12–22)\n\n" + scope.split("\n", 2)[2] + "\n" + appA + parts["appB"] +
       "\n## Appendix C — numerics, and where parity is not defined by the reference itself\n\n" + numerics)
open(sys.argv[2], "w").write(out)
print(len(out.splitlines()), "lines")
