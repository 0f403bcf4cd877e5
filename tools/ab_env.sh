#!/bin/bash
# A/B of an environment switch on the whole step inside one GPU box (alternating):  tools/ab_env.sh DA_FUSED_STEP=0 [bench args]
SW=$1; shift
run() { env $2 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "${@:3}" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), round(d['value']))"; }
run base X=1 "$@" || exit 1
run "$SW" "$SW" "$@" || exit 1
run base X=1 "$@" || exit 1
run "$SW" "$SW" "$@" || exit 1
run base X=1 "$@" || exit 1
