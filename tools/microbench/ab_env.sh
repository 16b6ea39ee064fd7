#!/bin/bash
# A/B the whole-step bench on ONE box between an env-var variant and the default: ab_env.sh VAR [bench args]
v=$1; shift
for r in 1 2 3; do
for s in "" 1; do
  if [ -n "$s" ]; then export $v=1; else unset $v; fi
  python bench.py "$@" 2>/dev/null | tail -1 > /tmp/b.json
  python - "$v=$s" <<PY
import json, sys
d = json.load(open("/tmp/b.json"))
print(f"{sys.argv[1]:28s} {d['value']:9.1f} img/s {d['ms_per_step']:7.3f} ms")
PY
done; done
