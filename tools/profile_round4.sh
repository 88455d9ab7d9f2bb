# Round 4 profiles (run on the GPU box from the repository root): rocprofv3 kernel traces with statistics, the two HBM-traffic PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate runs, --kernel-trace only beside them), one SQ / TCP / GRBM counter pass for the question "what binds
# k_msm_rows<0>: instruction issue or the gather" (VERDICT r3 item 3), and a kernel trace of the six-proofs-in-flight leg, for the NIZK
# headline (bench.py) and SNARK mode (tools/snark_probe.py).  The program itself follows `--` (python3 directly).
set -e
R=$PWD; O=$R/gpurun_out/r4p; mkdir -p $O && cd /tmp && export TMPDIR=/tmp
export OTTI_ARMED=0   # armed launches and the persistent sum-check tail wait for the host inside the kernel: off, so that every duration in the trace is the kernel alone
NIZK="--in-flight -1 --no-cpu-baseline --no-e2e --no-snark --no-sweep"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 10 --warmup 2 $NIZK > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 1 $NIZK > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 1 $NIZK > $O/pmc_write.json 2> $O/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 1 --warmup 1 $NIZK > $O/pmc_sq.json 2> $O/pmc_sq.err || echo "SQ pass failed"
rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $O/pmc_tcp -- python3 $R/bench.py --steps 1 --warmup 1 $NIZK > $O/pmc_tcp.json 2> $O/pmc_tcp.err || echo "TCP pass failed"
unset OTTI_ARMED
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_inflight -- python3 $R/bench.py --steps 4 --warmup 1 --in-flight 6 --no-cpu-baseline --no-e2e --no-snark --no-sweep > $O/bench_inflight_under_rocprof.json 2> $O/bench_inflight_under_rocprof.err
export OTTI_ARMED=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/snark_kt -- python3 $R/tools/snark_probe.py 20 5 > $O/snark_under_rocprof.txt 2>&1 || echo "snark kernel trace: exit status $?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/snark_pmc_fetch -- python3 $R/tools/snark_probe.py 20 1 > $O/snark_pmc_fetch.txt 2>&1 || echo "snark FETCH_SIZE pass: exit status $?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/snark_pmc_write -- python3 $R/tools/snark_probe.py 20 1 > $O/snark_pmc_write.txt 2>&1 || echo "snark WRITE_SIZE pass: exit status $?"
cd $R
find gpurun_out/r4p -name "*.csv" | sort
