for i in 1 2; do
for cfg in "1 1" "1 11" "0 1" "0 11"; do set -- $cfg
 env SSLCR_BN_ONE_LAUNCH=$1 SSLCR_BN_REPEAT=$2 python bench.py --steps 20 --warmup 5 --no-also --no-cpu-baseline --no-pmc --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one=$1 rep=$2', d['ms_per_step'])"
done; done
