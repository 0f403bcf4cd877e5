#!/bin/bash
# output-stage ablations (timing-only variants of k_edge_io.hip): kernel stats of bench.py per variant, one box
cd /tmp && export TMPDIR=/tmp
for v in base ot_NODRAW ot_NOMFMA ot_NOWRITE ot_ALL3; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_ot_$v; rm -rf $OUT; mkdir -p $OUT
  if [ $v = base ]; then unset DA_LIB; else export DA_LIB=$GRAFT_REPO_ROOT/diffalign_amd/libdiffalign_hip_$v.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-loop > $OUT/bench.log 2>&1 || { echo "$v failed"; tail -3 $OUT/bench.log; exit 1; }
  echo "== $v $(grep -h k_edge_out_tiles $OUT/b_kernel_stats.csv | cut -d, -f2-4 | head -1)"
done
