import sys,re,collections
d=collections.defaultdict(list)
for line in sys.stdin:
    m=re.match(r"\[rpvg_(amd|hip) trace\]\s+(.*?)\s+([0-9.]+) ms", line)
    if m: d[m.group(2)].append(float(m.group(3)))
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])/len(kv[1])):
    v2=v[1:] if len(v)>2 else v
    print("%-48s n=%2d mean %8.3f min %8.3f max %8.3f" % (k,len(v2),sum(v2)/len(v2),min(v2),max(v2)))
