#!/usr/bin/env python3
"""Combine the per-kernel FETCH_SIZE and WRITE_SIZE summaries (tools/pmc_summary.py output of two separate
rocprofv3 --pmc passes of `bench.py --no-cpu-baseline`) into profiles/<round>_pmc_traffic.json, keyed by the
kernel classes bench.py reports.  usage: pmc_to_json.py FETCH.csv WRITE.csv "fem2d L=7 p=1" > out.json"""
import csv
import json
import sys

CLASSES = {          # rocprofv3 kernel name fragment -> bench.py kernel class
    "front_start_kernel": "chol_front_start", "front_step_kernel": "chol_front_step", "front_step2_kernel": "chol_front_step", "front_single_kernel": "chol_front_single", "front_leaf_kernel": "chol_front_leaf",
    "backward_rect_kernel": "chol_backward_rect", "backward_kernel": "chol_backward",
    "spmv_kernel_t<4,": "apply_D", "spmv_kernel_t<8,": "hessian_assemble_plan", "spmv_kernel_t<16,": "restrict",
    "elop_assemble_kernel": "hessian_assemble", "gather_sum_kernel": "hessian_assemble_gather",
    "barrier_f1_kernel_t": "barrier_f1", "barrier_f2_kernel_t": "barrier_f2",
    "barrier_f0_kernel": "barrier_f0_unfused", "trial_f0_kernel": "barrier_f0", "barrier_f1_kernel": "barrier_f1", "barrier_f2_kernel": "barrier_f2",
}


def read(path):
    out = {}
    for r in csv.DictReader(open(path)):
        out[r["kernel"]] = (float(r["mean_value"]), int(r["dispatches"]))
    return out


fetch, write = read(sys.argv[1]), read(sys.argv[2])
kern = {}
for name, (fv, nd) in fetch.items():
    for frag, cls in CLASSES.items():
        if frag in name and not (frag == "backward_kernel" and "rect" in name):
            wv = write.get(name, (0.0, 0))[0]
            d = kern.setdefault(cls, dict(rocprof_name=name, dispatches=0, fetch_kib=0.0, write_kib=0.0))
            # several template instances (e.g. backward_kernel<256>, <1024>) fold into one class: dispatch-weighted mean
            tot = d["dispatches"] + nd
            d["fetch_kib"] = (d["fetch_kib"] * d["dispatches"] + fv * nd) / tot
            d["write_kib"] = (d["write_kib"] * d["dispatches"] + wv * nd) / tot
            d["dispatches"] = tot
            break
for d in kern.values():
    d["hbm_bytes_per_launch"] = (2.0 * d["fetch_kib"] + d["write_kib"]) * 1024.0
json.dump({
    "workload": sys.argv[3],
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --no-cpu-baseline",
    "units": "FETCH_SIZE/WRITE_SIZE in KiB per dispatch (mean); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: "
             "FETCH_SIZE reports half of a wide coalesced read, MI355X_MICROARCH.md HBM section)",
    "kernels": kern}, sys.stdout, indent=1)
