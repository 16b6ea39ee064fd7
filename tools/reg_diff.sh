#!/bin/bash
# registers and scratch of every kernel in csrc/*.hip, working tree against a git revision: tools/reg_diff.sh <rev> [file.hip ...]
# (an edit that costs a hot instance its last free register shows up here, not in the tests: round 6 lost 10 % of the dominant conv that way)
REV=${1:-HEAD}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/sslcr_regdiff; rm -rf $W; mkdir -p $W/old/ssl_cr_histo_amd/csrc $W/old/include $W/new/ssl_cr_histo_amd/csrc $W/new/include
FILES=${@:-$(cd $ROOT/ssl_cr_histo_amd/csrc && ls *.hip)}
for f in $(cd $ROOT/ssl_cr_histo_amd/csrc && ls *.hpp); do git -C $ROOT show $REV:ssl_cr_histo_amd/csrc/$f > $W/old/ssl_cr_histo_amd/csrc/$f 2>/dev/null; cp $ROOT/ssl_cr_histo_amd/csrc/$f $W/new/ssl_cr_histo_amd/csrc/; done
git -C $ROOT show $REV:include/sslcr.h > $W/old/include/sslcr.h; cp $ROOT/include/sslcr.h $W/new/include/
for f in $FILES; do
  git -C $ROOT show $REV:ssl_cr_histo_amd/csrc/$f > $W/old/ssl_cr_histo_amd/csrc/$f 2>/dev/null || continue
  cp $ROOT/ssl_cr_histo_amd/csrc/$f $W/new/ssl_cr_histo_amd/csrc/
  cmp -s $W/old/ssl_cr_histo_amd/csrc/$f $W/new/ssl_cr_histo_amd/csrc/$f && [ -z "$FORCE" ] && continue
  extra=""; case $f in conv_pp64.hip|stem.hip) extra="-fno-slp-vectorize";; esac
  for side in old new; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $extra -S --cuda-device-only -x hip $W/$side/ssl_cr_histo_amd/csrc/$f -o $W/$side.s 2>/dev/null & done; wait
  python3 - "$f" $W/old.s $W/new.s <<'PY'
import re, sys
def parse(p):
    s = open(p).read(); d = {}
    for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.vgpr_count:\s+(\d+)', s, re.S):
        ps = re.search(r'\.private_segment_fixed_size:\s+(\d+)', m.group(0))
        if 'kernel' in m.group(1): d[m.group(1)] = (int(m.group(3)), int(ps.group(1)) if ps else -1)
    return d
a, b = parse(sys.argv[2]), parse(sys.argv[3])
ch = [(k, a.get(k), b[k]) for k in b if a.get(k) != b[k]]
print(f"### {sys.argv[1]}: {len(b)} kernels, {len(ch)} changed (vgpr, scratch bytes)")
for k, o, n in ch: print("   ", k[-80:], o, "->", n)
PY
done
