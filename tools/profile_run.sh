# Run on the GPU box from the repository root: rocprofv3 kernel statistics of the
# benchmark's legs and the two PMC passes.  Output under gpurun_out/prof_$1/.
set -e
tag=$1
out=$PWD/gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rotated -- python3 bench.py --measure rotated --no-cpu-baseline > $out/rotated.json 2> $out/rotated.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/resident -- python3 bench.py --measure resident --no-cpu-baseline > $out/resident.json 2> $out/resident.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/step -- python3 bench.py --measure step --steps 2000 --warmup 200 > $out/step.json 2> $out/step.err
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 tools/pmc_driver.py > $out/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 tools/pmc_driver.py > $out/pmc_write.log 2>&1
python3 tools/collect_profiles.py $tag $out/pmc_fetch $out/pmc_write \
  "rotated=$out/rotated=python3 bench.py --measure rotated --no-cpu-baseline" \
  "resident=$out/resident=python3 bench.py --measure resident --no-cpu-baseline" \
  "step=$out/step=python3 bench.py --measure step --steps 2000 --warmup 200"
cp profiles/${tag}_* profiles/traffic.json $out/
# the raw traces are large: keep the summaries only
rm -rf $out/rotated $out/resident $out/step $out/pmc_fetch $out/pmc_write
