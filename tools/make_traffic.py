#!/usr/bin/env python3
"""profiles/traffic_rNN.json from a `tools/pmc_summary.py` summary of a `rocprofv3 --pmc` pass over bench.py: the per-launch HBM
traffic and VALU / SALU instruction counts of the scan kernel that `bench.py` quotes as `roofline.traffic` /
`valu_wave_instructions_per_launch`, TOGETHER WITH the sha256 of the kernel sources they were measured on
(`bench.kernel_source_hash()`): bench.py reports the counters only while that hash still matches the tree.

    python tools/make_traffic.py PMC_SUMMARY.json [--clock-held GHZ] [--prev profiles/traffic_r02.json] > profiles/traffic_r03.json

Bytes = FETCH_SIZE (KB) x 1024 x 2: MI355X_MICROARCH.md, HBM section -- FETCH_SIZE counts 64 B per 128-B request for 16-byte-per-lane
streams (calibrated in r01 on gather_rows: 516 MB read -> 544 MB counted).  Schedules the pass did not run (query-major, wave-level
bucket-major: kernels unchanged since r01) are carried over from --prev and labelled so."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pmc")
    ap.add_argument("--clock-held", type=float, default=None)
    ap.add_argument("--prev", default=os.path.join(ROOT, "profiles", "traffic_r02.json"))
    ap.add_argument("--method", default="")
    args = ap.parse_args()
    import bench
    pmc = json.load(open(args.pmc))
    prev = json.load(open(args.prev))
    name = next(k for k in pmc if "bscan3_kernel<0" in k)
    c = pmc[name]
    out = {
        "workload": prev["workload"],
        "kernel_source_sha256": bench.kernel_source_hash(),
        "kernel_sources": list(bench.KERNEL_SOURCES),
        "method": args.method or prev["method"],
        "algorithmic_bytes_per_launch": prev["algorithmic_bytes_per_launch"],
        "traffic_bytes_per_launch": {"2": c["FETCH_SIZE"] * 1024 * 2},
        "valu_wave_instructions_per_launch": {"2": c["SQ_INSTS_VALU"]},
        "salu_wave_instructions_per_launch": {"2": c["SQ_INSTS_SALU"]},
        "kernels": {"2": name + " (this round's pass, tools/final_measure.sh)"},
        "carried_over_from_" + os.path.basename(args.prev): {"traffic_bytes_per_launch": {k: v for k, v in prev["traffic_bytes_per_launch"].items() if k != "2"},
                                                              "note": "schedules 0 and 1: kernels unchanged since the pass that measured them; not reported by bench.py (its guard wants this round's hash)"},
    }
    if args.clock_held:
        out["clock_held_GHz"] = {"2": args.clock_held}
        out["note_clock"] = "in-kernel shader clock of the tiled kernel on this workload (tools/scan_clock.py on a -DNLSH_SCAN_TRACE -DNLSH_SCAN_TRACE_CLOCK build), median over workgroups longer than 20 us"
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
