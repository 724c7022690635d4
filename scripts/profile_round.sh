#!/bin/bash
# Round profile on the GPU box: kernel-trace stats, HBM traffic counters (separate --pmc passes, no other trace domains),
# and the bench line.  Everything lands in gpurun_out/<tag>_*; copy what should be judged into profiles/.
#   gpurun -- bash scripts/profile_round.sh r01g
tag=${1:-round}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { timeout 300 "$@" </dev/null; }
run rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python $root/scripts/prof_run.py kitti00 5 > $out/${tag}_stats.log 2>&1
cp "$(find $out/${tag}_stats -name '*kernel_stats.csv' | head -1)" $out/${tag}_kitti00_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  run rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/${tag}_pmc_$c -- python $root/scripts/prof_run.py kitti00 1 > $out/${tag}_pmc_$c.log 2>&1
done
cd $root
python scripts/pmc_traffic.py $out/${tag}_pmc_FETCH_SIZE $out/${tag}_pmc_WRITE_SIZE > $out/${tag}_kitti00_pmc_traffic.json
cp $out/${tag}_kitti00_pmc_traffic.json profiles/        # the bench line cites the traffic measured in this very run
# the large shapes (BASELINE configs[2] and [4] on one GPU): kernel stats + traffic of one run each, cited by the bench line's shapes.*.roofline
for shp in s2m g4m; do
  (cd /tmp && run rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_${shp}_stats -- python $root/scripts/prof_run.py $shp 2 > $out/${tag}_${shp}_stats.log 2>&1)
  f=$(find $out/${tag}_${shp}_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $out/${tag}_${shp}_kernel_stats.csv && cp "$f" profiles/${tag}_${shp}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && run rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/${tag}_${shp}_pmc_$c -- python $root/scripts/prof_run.py $shp 0 > $out/${tag}_${shp}_pmc_$c.log 2>&1)
  done
  python scripts/pmc_traffic.py $out/${tag}_${shp}_pmc_FETCH_SIZE $out/${tag}_${shp}_pmc_WRITE_SIZE > $out/${tag}_${shp}_pmc_traffic.json && cp $out/${tag}_${shp}_pmc_traffic.json profiles/
  rm -rf $out/${tag}_${shp}_pmc_FETCH_SIZE $out/${tag}_${shp}_pmc_WRITE_SIZE $out/${tag}_${shp}_stats      # (raw counter dumps of the large shapes: hundreds of MB)
done
cp $out/${tag}_kitti00_kernel_stats.csv profiles/
python bench.py --steps 50 --warmup 10 > $out/${tag}_bench.json 2> $out/${tag}_bench.err
# the same command under the tracer (main leg only): its per-kernel averages must agree with the line's HIP-event times
(cd /tmp && run rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_bench_stats -- python $root/bench.py --steps 50 --warmup 10 --no-shapes --no-end-to-end > $out/${tag}_bench_stats.log 2>&1)
f=$(find $out/${tag}_bench_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $out/${tag}_bench_py_kernel_stats.csv
tail -c 600 $out/${tag}_bench.json
