#!/usr/bin/env python3
"""gpurun_out/r03_pmc/ (`tools/gpu_r06.sh pmc ...`; rounds 3-5: the collect_r0N.sh / gpu_r05.sh scripts of their time, in git history) -> profiles/pmc_constants.json + the CSV summaries under profiles/rNN/.

For every workload: FETCH_SIZE and WRITE_SIZE (KiB) of its roofline kernels, averaged per dispatch of the bench's launches and summed over
the kernels of the group, with the units one launch processes and THE COMMIT the library was built from — bench.py quotes that commit
beside `roofline.traffic`, so a constant that has gone stale behind a changed kernel shows."""
import csv, glob, json, os, subprocess, sys, collections, shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r03"
SRC = os.path.join(ROOT, "gpurun_out", "%s_pmc" % ROUND)
GROUPS = {   # name -> (units per launch, kernel-name fragments of its roofline kernels[, the run the counters come from when it is not `name`])
    "linear": (400000, ["gssw_fill_kernel"]),
    "banded": (100000, ["banded_fill_kernel"]),
    "gapless": (1000000, ["gapless_search_kernel", "gapless_rules_kernel", "gapless_kernel("]),
    "wfa": (500000, ["wfa_kernel", "wfa_wave_kernel"]),
    "xband": (200000, ["xdrop_band_pk_kernel", "xdrop_band_kernel", "xdrop_band_walk_kernel"]),
    # configs[2], one batch of 1 M reads in one context: the extension kernels of the stage, and its seeding kernels on their own
    "config2": (1000000, ["gapless_search_kernel", "gapless_rules_kernel", "gapless_kernel("], "config2"),
    "minimizer": (1000000, ["minimizer_kernel", "minimizer_gather_kernel"], "config2"),
    # configs[4]: the WFA launches of the chain stage, per read (4 000 reads per launch)
    "longread": (4000, ["wfa_wave_kernel", "wfa_kernel"], "longread"),
    # configs[3] slice: the rescue half's kernels (fills and tracebacks of its three rounds) per PAIR of the batch; a step launches them a varying number of
    # times, so the total is divided by the stage runs of the profiled command (warmup 1 + steps 2 + the untimed run that fetches the ops = 4)
    "rescue": (250000, ["gssw_fill_kernel", "gssw_walk"], "paired", 4),
}


def per_dispatch(path, counter, frags, runs=None):
    """the group's counter total per BATCH of the bench: every kernel's total divided by the batches run = the fewest dispatches any kernel of
    the group has (a kernel launched several times per batch — the seeding kernel's four slices — counts with all of them)"""
    tot = collections.defaultdict(float); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        if not any(f in k for f in frags):
            continue
        tot[k] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    batches = runs if runs else (min(len(disp[k]) for k in tot) if tot else 1)
    return sum(tot[k] for k in tot) / batches, {k.split("(")[0]: len(disp[k]) for k in tot}


def main():
    commit = subprocess.check_output(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT).decode().strip()
    out_path = os.path.join(ROOT, "profiles", "pmc_constants.json")
    table = json.load(open(out_path)) if os.path.exists(out_path) else {}
    dst = os.path.join(ROOT, "profiles", ROUND)
    os.makedirs(dst, exist_ok=True)
    only = set(sys.argv[2:])        # e.g. `pmc_constants.py r03 xband`: the other workloads keep their figures and the commit those were measured at
    for name, spec in GROUPS.items():
        units, frags = spec[0], spec[1]; run = spec[2] if len(spec) > 2 else name
        if only and name not in only:
            continue
        vals = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            found = glob.glob(os.path.join(SRC, "%s_%s" % (counter, run), "**", "*counter_collection.csv"), recursive=True)
            if not found:
                print("missing", counter, name); break
            vals[counter], dispatches = per_dispatch(found[0], counter, frags, spec[3] if len(spec) > 3 else None)
            shutil.copy(found[0], os.path.join(dst, "pmc_%s_%s_%s.csv" % (counter.split("_")[0].lower(), run, ROUND)))
        else:
            table[name] = {"fetch_kib_per_launch": vals["FETCH_SIZE"], "write_kib_per_launch": vals["WRITE_SIZE"], "units_per_launch": units,
                           "kernels": frags, "dispatches_averaged": dispatches, "commit": commit,
                           "files": ["profiles/%s/pmc_fetch_%s_%s.csv" % (ROUND, run, ROUND), "profiles/%s/pmc_write_%s_%s.csv" % (ROUND, run, ROUND)],
                           "bytes_per_unit": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024 / units}
            print(name, table[name]["bytes_per_unit"], "B/unit", dispatches)
        st = glob.glob(os.path.join(SRC, "stats_%s" % run, "**", "*kernel_stats.csv"), recursive=True)
        if st:
            shutil.copy(st[0], os.path.join(dst, "kernel_stats_%s_%s.csv" % (run, ROUND)))
    json.dump(table, open(out_path, "w"), indent=1, sort_keys=True)
    print("->", out_path)


if __name__ == "__main__":
    main()
