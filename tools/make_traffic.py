#!/usr/bin/env python3
"""profiles/traffic_rNN.json from `rocprofv3 --pmc` passes over bench.py: per workload, the per-launch HBM traffic and VALU / SALU
instruction counts of the scan kernel that `bench.py` quotes as `roofline.traffic` / `valu_wave_instructions_per_launch`, TOGETHER
WITH the sha256 of the kernel sources they were measured on (`bench.kernel_source_hash()`): bench.py reports the counters only while
that hash still matches the tree, and only for the entry whose workload shape, schedule and row window equal the run's.

    python tools/make_traffic.py --entry PMC_SUMMARY.json:BENCH_LINE.json[:CLOCK_GHZ] [--entry ...] > profiles/traffic_r06.json

PMC_SUMMARY.json = tools/pmc_summary.py over the pass; BENCH_LINE.json = the line bench.py printed IN that pass (its config.shape and
config.traffic_key name the entry).  Bytes = FETCH_SIZE (KB) x 1024 x 2: MI355X_MICROARCH.md, HBM section -- FETCH_SIZE counts 64 B
per 128-B request for 16-byte-per-lane streams (calibrated in r01 on gather_rows: 516 MB read -> 544 MB counted)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--entry", action="append", required=True)
    ap.add_argument("--origin", action="append", default=[], metavar="TRAFFIC_KEY:EA_REQUESTS.json:FLUSH_EXPERIMENT.jsonl",
                    help="r06 (VERDICT r05 item 3): where an entry's bytes come from, as far as this pool can tell -- the fabric-side read requests of the launch "
                         "(TCC_EA0_RDREQ / _DRAM / _32B via tools/pmc_summary.py) and the same launch with the Infinity Cache flushed between batches "
                         "(tools/order_alternation.py); there is no counter behind the fabric, so `dram_bytes_per_launch` itself stays null")
    ap.add_argument("--method", default="rocprofv3 --pmc FETCH_SIZE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU -- "
                                         "python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --query-chunks 1 [workload flags] (tools/final_measure.sh); per-launch average "
                                         "of the scan kernel (tools/pmc_summary.py); bytes = FETCH_SIZE(KB) * 1024 * 2")
    args = ap.parse_args()
    import bench
    out = {"kernel_source_sha256": bench.kernel_source_hash(), "kernel_sources": list(bench.KERNEL_SOURCES), "method": args.method, "entries": {}}
    for spec in args.entry:
        parts = spec.split(":")
        pmc = json.load(open(parts[0]))
        line = json.loads([ln for ln in open(parts[1]).read().splitlines() if ln.startswith("{")][-1])
        kern = {0: "scan_kernel", 1: "bscan2_kernel", 2: "bscan3_kernel"}[line["config"]["shape"]["algo"]]
        name = max((k for k in pmc if kern in k), key=lambda k: pmc[k].get("SQ_INSTS_VALU", 0))
        c = pmc[name]
        ent = {"workload": line["config"]["shape"], "kernel": name, "traffic_bytes_per_launch": c["FETCH_SIZE"] * 1024 * 2,
               "valu_wave_instructions_per_launch": c.get("SQ_INSTS_VALU"), "salu_wave_instructions_per_launch": c.get("SQ_INSTS_SALU"),
               "sq_wave_cycles": c.get("SQ_WAVE_CYCLES"), "sq_wait_any": c.get("SQ_WAIT_ANY"),
               "distinct_candidate_row_bytes_per_launch": line["roofline"].get("distinct_candidate_row_bytes_per_launch"),
               "algorithmic_bytes_per_launch": line["roofline"]["algorithmic_bytes_per_launch"]}
        if len(parts) > 2:
            ent["clock_held_GHz"] = float(parts[2])
            ent["note_clock"] = "in-kernel shader clock of the tiled kernel on this workload (tools/scan_clock.py on a -DNLSH_SCAN_TRACE -DNLSH_SCAN_TRACE_CLOCK build), median over workgroups longer than 20 us"
        out["entries"][line["config"]["traffic_key"]] = ent
    for spec in args.origin:
        key, ea_file, flush_file = spec.rsplit(":", 2)      # the traffic key itself holds colons
        ea = json.load(open(ea_file))
        ea = ea[next(iter(ea))]
        wl = {"glove:manifold:exact": "glove", "sift1m:clusters:exact": "clusters", "sift1m:manifold:exact": "sift1m"}[key]
        fl = next(json.loads(ln) for ln in open(flush_file) if ln.startswith("{") and json.loads(ln).get("workload") == wl)
        out["entries"][key]["origin"] = {
            "dram_bytes_per_launch": None,
            "why_null": "rocprofv3 --list-avail on this pool has no counter behind the fabric (no UMC / DF / MALL block): Infinity-Cache hits cannot be told from DRAM reads",
            "fabric_read_requests_per_launch": ea.get("TCC_EA0_RDREQ_sum"), "of_which_routed_to_local_dram": ea.get("TCC_EA0_RDREQ_DRAM_sum"),
            "of_which_32_byte": ea.get("TCC_EA0_RDREQ_32B_sum"),
            "scan_ms_same_order_every_batch": [fl["same_order_0"][0], fl["same_order_1"][0]],
            "scan_ms_infinity_cache_flushed_between_batches": fl["same_order_cache_flushed_between_batches"][0],
            "reading": "routing, not residency: every read request of the launch goes towards local DRAM; with a 512-MiB write between batches (nothing of the previous batch left on "
                       "the die) the same launch is this much slower, i.e. consecutive batches already get a large share of their rows from the Infinity Cache"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
