"""Development aid: instruction census of one kernel in a `hipcc -S` listing.
    python scripts/isa_census.py file.s kernel-name-fragment"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
for name in sys.argv[2:]:
    m = re.search(r"^(\S*%s\S*):" % re.escape(name), s, re.M)
    i = m.end()
    k = s.index(".Lfunc_end", i)
    c = collections.Counter()
    for line in s[i:k].split("\n"):
        line = line.strip()
        if not line or line[0] in ";." or line.endswith(":"):
            continue
        c[line.split()[0]] += 1
    valu = sum(v for k2, v in c.items() if k2.startswith("v_"))
    print(name, "total", sum(c.values()), "VALU", valu)
    print("   ", ", ".join(f"{k2}:{v}" for k2, v in c.most_common(50)))
