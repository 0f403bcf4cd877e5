#!/bin/bash
# A/B of environment settings on the whole step inside one GPU box: tools/ab_env_args.sh "VAR=a VAR2=b" "VAR=c" -- bench args
# runs base (no setting), each setting, base again.
sets=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do sets+=("$1"); shift; done; shift
run() { n=$1; shift; timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-loop "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', round(d['ms_per_step'],3))"; }
run base "$@" || exit 1
for s in "${sets[@]}"; do ( export $s; run "$s" "$@" ) || exit 1; done
run base "$@"
