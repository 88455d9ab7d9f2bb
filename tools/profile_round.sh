set -e
R=$PWD; mkdir -p gpurun_out/r2p && cd /tmp && export TMPDIR=/tmp
export OTTI_ARMED=0   # armed launches wait for the host inside the kernel: switched off so that every kernel duration in the trace is the kernel alone
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2p/kt -- python3 $R/bench.py --steps 10 --warmup 2 --in-flight -1 --no-cpu-baseline --no-e2e --no-snark > $R/gpurun_out/r2p/bench_under_rocprof.json 2> $R/gpurun_out/r2p/bench_under_rocprof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r2p/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --in-flight -1 --no-cpu-baseline --no-e2e --no-snark > $R/gpurun_out/r2p/pmc_fetch.json 2> $R/gpurun_out/r2p/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r2p/pmc_write -- python3 $R/bench.py --steps 1 --warmup 1 --in-flight -1 --no-cpu-baseline --no-e2e --no-snark > $R/gpurun_out/r2p/pmc_write.json 2> $R/gpurun_out/r2p/pmc_write.err
cd $R
find gpurun_out/r2p -name "*.csv" | head -20
