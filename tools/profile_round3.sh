# Round 3 profiles (run on the GPU box from the repository root): rocprofv3 kernel traces with statistics, and the two PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate runs, --kernel-trace only beside them), for the NIZK headline (bench.py) and for SNARK mode
# (tools/snark_probe.py).  The program itself follows `--` (python3 directly: the profiler's preloaded library initialises the GPU).
set -e
R=$PWD; O=$R/gpurun_out/r3p; mkdir -p $O && cd /tmp && export TMPDIR=/tmp
export OTTI_ARMED=0   # armed launches and the persistent sum-check tail wait for the host inside the kernel: off, so that every duration in the trace is the kernel alone
NIZK="--steps 10 --warmup 2 --in-flight -1 --no-cpu-baseline --no-e2e --no-snark --no-sweep"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py $NIZK > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --in-flight -1 --no-cpu-baseline --no-e2e --no-snark --no-sweep > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 1 --in-flight -1 --no-cpu-baseline --no-e2e --no-snark --no-sweep > $O/pmc_write.json 2> $O/pmc_write.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/snark_kt -- python3 $R/tools/snark_probe.py 20 5 > $O/snark_under_rocprof.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/snark_pmc_fetch -- python3 $R/tools/snark_probe.py 20 1 > $O/snark_pmc_fetch.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/snark_pmc_write -- python3 $R/tools/snark_probe.py 20 1 > $O/snark_pmc_write.txt 2>&1
cd $R
find gpurun_out/r3p -name "*.csv" | sort
