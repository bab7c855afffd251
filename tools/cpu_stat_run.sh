#!/bin/bash
# runs the given command and prints the cgroup's user/system CPU seconds it used
s0=$(grep -E "^(user|system)_usec" /sys/fs/cgroup/cpu.stat | awk '{print $2}' | tr '\n' ' ')
"$@"
s1=$(grep -E "^(user|system)_usec" /sys/fs/cgroup/cpu.stat | awk '{print $2}' | tr '\n' ' ')
python3 - <<PY
a=[int(x) for x in "$s0".split()]; b=[int(x) for x in "$s1".split()]
print("cpu seconds: user %.1f system %.1f" % ((b[0]-a[0])/1e6, (b[1]-a[1])/1e6))
PY
