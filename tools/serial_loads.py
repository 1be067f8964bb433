"""Where a listing waits for single memory loads: every vector-memory load that is followed, within four instructions and
before another load, by `s_waitcnt vmcnt(0)` -- a load that was issued alone and waited for on the spot -- counted by the
source line it comes from.  Loads behind a guard (`i < n ? p[i] : 0`) or behind a branch on a uniform value that is tested
inside an unrolled loop end up like this: four "loads in flight" become four memory latencies in a row.

    REGCHECK_FLAGS=-gline-tables-only tools/regcheck.sh 512 10 1 0 true && python tools/serial_loads.py [/tmp/regcheck/one.s]
"""
import collections
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/regcheck/one.s"
lines = open(path).read().split("\n")
files, cur, locs = {}, None, []
for l in lines:
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[m.group(1)] = (m.group(3) or m.group(2)).split("/")[-1]
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(m.group(1), m.group(1)), int(m.group(2)))
    locs.append(cur)


def is_load(l):
    return re.search(r"^\s+(global|flat|buffer|scratch)_load", l)


code = [(i, l) for i, l in enumerate(lines) if re.match(r"^\s+[a-z]", l) and not l.strip().startswith(".")]
hits = collections.Counter()
for k, (i, l) in enumerate(code):
    if is_load(l):
        for j in range(k + 1, min(k + 5, len(code))):
            nxt = code[j][1]
            if "s_waitcnt" in nxt and "vmcnt(0)" in nxt:
                hits[locs[i]] += 1
                break
            if is_load(nxt):
                break
if not any(locs):
    sys.exit("no .loc directives: compile with -gline-tables-only (REGCHECK_FLAGS)")
for (f, ln), c in sorted(hits.items(), key=lambda x: -x[1])[:30]:
    print(f"{c:4d}  {f}:{ln}")
