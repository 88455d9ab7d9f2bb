"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter_collection.csv each) into profiles/<round>_pmc_traffic.json:
per kernel, the launch with the largest FETCH_SIZE + WRITE_SIZE, and its HBM traffic in bytes.
Counter unit: KiB.  gfx950 correction (MI355X guide, HBM / rocprofv3 section): FETCH_SIZE reports half of the bytes of wide coalesced
reads, so it is doubled; WRITE_SIZE is taken as is.
usage: pmc_summary.py <fetch.csv> <write.csv> <out.json> [msm_window_bits [log2_constraints]]"""
import csv
import json
import sys

KERNELS = ("k_msm_rows<0>", "k_msm_small", "k_msm_rows<2>", "k_sc_cubic3_fold_eval", "k_sc_quad_fold_eval", "k_sc_cubic3_eval", "k_sc_quad_eval", "k_spmv3_light", "k_spmv3_light<true>", "k_spmv3_light<false>", "k_spmv3_quad<true>", "k_spmv3_quad<false>", "k_spmv3_heavy_seg<true>", "k_eq_expand",
           "k_poly_bound_slab", "k_gather_strided",
           # SNARK mode (k_snark.hip) and the verifier's kernels
           "k_pc_round<true>", "k_pc_round<false>", "k_pc_tail", "k_prod_layer", "k_hash_ops", "k_hash_mem", "k_gather", "k_dot_many", "k_sum3", "k_msm_var", "k_decode_niels")


def load(path, counter):
    rows = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        rows[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]))
    return rows


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in KERNELS:
        best = None
        for d, (name, f) in fetch.items():
            if (k + "(") not in name:
                continue
            w = write.get(d, (name, 0.0))
            w = w[1] if ((k + "(") in w[0]) else 0.0          # the two passes dispatch the same sequence
            if best is None or f + w > best[0] + best[1]:
                best = (f, w)
        if best:
            out[k] = {"largest_launch_FETCH_SIZE_KiB": best[0], "largest_launch_WRITE_SIZE_KiB": best[1],
                      "traffic_bytes_corrected": int((2 * best[0] + best[1]) * 1024)}
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over the same command (tools/profile_round4.sh (round 3: profile_round3.sh): bench.py for the NIZK kernels, "
                       "tools/snark_probe.py for SNARK mode; 2^20; window width in msm_window_bits). Unit of the counters: KiB. gfx950 correction (MI355X guide): FETCH_SIZE reports half of the "
                       "bytes of wide coalesced 16-B-per-lane reads, so it is doubled; for the scattered 16-B loads of the window-table gathers that "
                       "factor is not calibrated and the corrected figure is an upper estimate.",
               "msm_window_bits": int(sys.argv[4]) if len(sys.argv) > 4 else 12, "log2_constraints": int(sys.argv[5]) if len(sys.argv) > 5 else 20,
               "kernels": out}, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
