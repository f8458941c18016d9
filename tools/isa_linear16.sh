#!/bin/bash
# dev tool: register / wait summary of every k_linear16 instantiation (what tests/test_host_logic.py asserts)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -o /tmp/k_linear16.s "$(dirname "$0")/../danbo-pytorch_amd/csrc/k_linear16.hip" 2>&1 | grep -v warning
python3 - <<'PY'
import re, collections
t=open('/tmp/k_linear16.s').read()
meta={m.group(1):(m.group(2),m.group(3)) for m in re.finditer(r"\.name:\s+(\S*k_linear16I\S*)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", t)}
high = re.compile(r"\bv(24\d|25[0-5])\b|v\[(24\d|25[0-5]):")
for name in re.findall(r"^(_ZN5danbo10k_linear16I\S+):", t, re.M):
    body=t[t.index(name+":"):]; body=body[:body.index(".Lfunc_end")].split("\n")
    touching=[l.strip() for l in body if high.search(l)]
    waits=collections.Counter(re.search(r"vmcnt\(\d+\)", l).group(0) for l in body if "s_waitcnt" in l and "vmcnt" in l)
    print(name[22:42], 'scratch', meta[name][0], 'touching v240+:', len(touching), dict(waits))
PY
