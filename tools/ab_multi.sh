run() { env $2 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "${@:3}" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), round(d['value']))"; }
for v in X=1 DA_TAIL_CHUNKS=4 DA_TAIL_CHUNKS=8 DA_TAIL_CHUNKS=16 DA_STATS_TAIL=0 X=1; do run $v $v "$@" || exit 1; done
