#!/bin/bash
# A/B grid inside one box: {base lib, variant lib} x {env off, env on}:  tools/ab_grid.sh VARIANT ENV=1 [bench args]
V=$1; SW=$2; shift; shift
run() { env $2 $3 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "${@:4}" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), round(d['value']))"; }
for rep in 1 2; do
run "base" X=1 Y=1 "$@" || exit 1
run "base+$SW" $SW Y=1 "$@" || exit 1
run "$V" DA_LIB=$PWD/diffalign_amd/libdiffalign_hip_$V.so Y=1 "$@" || exit 1
run "$V+$SW" DA_LIB=$PWD/diffalign_amd/libdiffalign_hip_$V.so $SW "$@" || exit 1
done
