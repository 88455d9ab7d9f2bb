# PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, --kernel-trace only beside them) and a kernel trace with statistics over the
# compiler-like workload (bench.py --dist compiler): what the two sparse products really move.  Run on the GPU box from the repository root.
set -e
R=$PWD; O=$R/gpurun_out/r3c; mkdir -p $O && cd /tmp && export TMPDIR=/tmp
export OTTI_ARMED=0
F="--dist compiler --in-flight -1 --no-cpu-baseline --no-e2e --no-snark --no-sweep"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py $F --steps 10 --warmup 2 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py $F --steps 1 --warmup 1 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py $F --steps 1 --warmup 1 > $O/pmc_write.json 2> $O/pmc_write.err
cd $R
find gpurun_out/r3c -name "*.csv" | sort
